#!/usr/bin/env python3
"""Writes tests/golden/ref_f1f2_1080p.npz -- the one golden NUMBER the reference holds for the hot path.

Runs in the build container only (needs /root/reference and PIL); the fixture travels, the reference does not.

The reference's noise-filter study decodes two 1920x1080 test photographs it ships,
tests/noise_filter_benchmark/f1.jpg and f2.jpg, and counts the bytes whose difference exceeds the
threshold (tests/noise_filter_benchmark/v2.cu:106-114 getCountDifference: `abs(orig[i]-mod[i]) > 20` over
H*W*3 bytes of the BGR `Mat`s, called at v2.cu:215).  Its report states the result:
REPORT/report.tex:2594 "The number of pixels changed, if no filter is applied, is 369350, that is 5.93%".
The table that follows (report.tex:2601-2611) gives the share still changed after both frames went through
the K x K filter: mean K=3 3.37 %, Gaussian K=3 sigma=1 3.58 %.

The fixture holds the two decoded frames (PIL decode, RGB -> BGR = OpenCV's imread order) and
  count_gt20      369350  -- the reference-held expected count (strict >; `>=` gives 413893 and must not match)
  report_pct_*            -- the report's table entries
  resid_mean3 / resid_gauss3_s1 -- what the oracle's 3x3 filter leaves on this pair (our restatement's numbers:
                            they agree with the report's table to two digits; the reference's own kernel leaves
                            uninitialised halo bytes, v2.cu:50-54, so its third digit is not reproducible).
  resid_intacc_*          -- the same through the reference's CPU statement of the filter
                            (tests/noise_filter_benchmark/cpu.cu:72-98, int accumulator): 3.39 / 3.59 %; the
                            report's figures lie between the two statements.
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

REF = "/root/reference/tests/noise_filter_benchmark"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_f1f2_1080p.npz")
REPORT_COUNT = 369350          # REPORT/report.tex:2594
REPORT_PCT_MEAN3 = 3.37        # REPORT/report.tex:2603
REPORT_PCT_GAUSS3_S1 = 3.58    # REPORT/report.tex:2608


def decode_bgr(path):
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


def main():
    po.build()
    f1, f2 = decode_bgr(os.path.join(REF, "f1.jpg")), decode_bgr(os.path.join(REF, "f2.jpg"))
    assert f1.shape == f2.shape == (1080, 1920, 3)
    d = np.abs(f1.astype(np.int32) - f2.astype(np.int32))
    gt, ge = int((d > 20).sum()), int((d >= 20).sum())
    assert gt == REPORT_COUNT, (gt, "decoder differs from the one the report's number was taken with")
    resid = {}
    for name, k in (("mean3", np.full(9, np.float32(1.0 / 9), np.float32)),    # v2.cu:116-124, K = 3
                    ("gauss3_s1", po.gaussian_kernel(3, 1.0))):                   # v2.cu main, SIGMA = 1
        a, b = po.conv3x3(f1, 1920, 1080, k), po.conv3x3(f2, 1920, 1080, k)
        resid[name] = int((np.abs(a.astype(np.int32) - b.astype(np.int32)) > 20).sum())
        # the reference's CPU statement of the same filter (cpu.cu:72-98: int accumulator, truncation per tap)
        ia, ib = po.conv3x3_intacc(f1, 1920, 1080, k), po.conv3x3_intacc(f2, 1920, 1080, k)
        resid["intacc_" + name] = int((np.abs(ia - ib) > 20).sum())
    np.savez_compressed(OUT, f1=f1, f2=f2, count_gt20=gt, count_ge20=ge,
                        report_pct_mean3=REPORT_PCT_MEAN3, report_pct_gauss3_s1=REPORT_PCT_GAUSS3_S1,
                        **{"resid_" + k: v for k, v in resid.items()})
    n = f1.size
    print(f"{OUT}: {os.path.getsize(OUT)} bytes; count {gt} (>=: {ge}); residual mean3 {resid['mean3']} "
          f"= {100.0 * resid['mean3'] / n:.2f} % (report {REPORT_PCT_MEAN3}), gauss3 sigma 1 "
          f"{resid['gauss3_s1']} = {100.0 * resid['gauss3_s1'] / n:.2f} % (report {REPORT_PCT_GAUSS3_S1}); "
          f"int-accumulator statement {resid['intacc_mean3']} / {resid['intacc_gauss3_s1']}")


if __name__ == "__main__":
    main()
