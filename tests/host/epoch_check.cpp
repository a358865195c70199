// tests/host/epoch_check.cpp -- host-side check of the index kernel's launch tags (csrc/internal.h, next_scan_epoch): the tag
// never is 0, wraps at 2^33 with the "clear the totals" signal, and a 31-bit total and a 33-bit tag share a 64-bit word.
// Built and run by tests/test_cabi.py with plain g++ (no GPU).
#include "internal.h"
#include <cstdio>
int main() {
    using namespace mi355;
    uint64_t e = 0;
    bool w = next_scan_epoch(e);
    if (w || e != 1) return 1;
    e = kEpochWrap - 2;
    w = next_scan_epoch(e);
    if (w || e != kEpochWrap - 1) return 2;
    w = next_scan_epoch(e);
    if (!w || e != 1) return 3;
    // the word: a total of 2^31 - 1 and the largest tag fit 64 bits without touching each other
    const uint64_t word = (uint64_t)((1u << kTotalBits) - 1u) | ((kEpochWrap - 1) << kTotalBits);
    if ((word >> kTotalBits) != kEpochWrap - 1 || (word & ((1ull << kTotalBits) - 1)) != (1u << kTotalBits) - 1u) return 4;
    printf("ok %llu\n", (unsigned long long)kEpochWrap);
    return 0;
}
