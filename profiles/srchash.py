#!/usr/bin/env python3
"""SHA-256 over the SOURCES libmi355diff.so is built from (csrc/*.hip, csrc/*.h, csrc/Makefile, include/mi355diff.h), in a
fixed order.  The library file itself differs between build directories (the compiler embeds a path-derived module id),
so counter summaries are keyed to this hash as well as to the library's: bench.py reports roofline.traffic when either
matches what it runs (profiles/pmc_summary.json).  Prints the hash when run."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def src_sha256(root=ROOT):
    csrc = os.path.join(root, "cudavideostream_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")))
    files += [os.path.join(csrc, "Makefile"), os.path.join(root, "include", "mi355diff.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(src_sha256())
