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
  table_*                 -- the WHOLE table of report.tex:2601-2611, nine rows: kind (0 mean, 1 Gaussian), K,
                            sigma, the report's percentage, and what the oracle's K x K filter (v2.cu:36-80
                            restated, ora_conv_kxk) leaves; every row agrees with the report to 0.03 points.
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
    # report.tex:2601-2611: (kind, K, sigma, percent)
    table = [(0, 3, 0, 3.37), (0, 5, 0, 2.31), (0, 7, 0, 1.66), (0, 9, 0, 1.24),
             (1, 3, 1, 3.58), (1, 4, 2, 2.87), (1, 5, 3, 2.37), (1, 6, 5, 1.98), (1, 7, 8, 1.66)]
    t_resid = []
    for kind, K, sigma, pct in table:
        k = po.mean_kernel(K) if kind == 0 else po.gaussian_kernel(K, float(sigma))
        a, b = po.conv_kxk(f1, 1920, 1080, k), po.conv_kxk(f2, 1920, 1080, k)
        r = int((np.abs(a.astype(np.int32) - b.astype(np.int32)) > 20).sum())
        assert abs(100.0 * r / f1.size - pct) < 0.05, (kind, K, sigma, r)
        t_resid.append(r)
        print(f"  table row {'mean' if kind == 0 else 'gauss'} K={K} sigma={sigma}: {r} = {100.0 * r / f1.size:.3f} % (report {pct})")
    assert t_resid[0] == resid["mean3"] and t_resid[4] == resid["gauss3_s1"]
    resid_tab = dict(table_kind=np.array([t[0] for t in table], np.int32), table_K=np.array([t[1] for t in table], np.int32),
                     table_sigma=np.array([t[2] for t in table], np.float32),
                     table_report_pct=np.array([t[3] for t in table], np.float64), table_resid=np.array(t_resid, np.int64))
    resid.update(resid_tab)
    np.savez_compressed(OUT, f1=f1, f2=f2, count_gt20=gt, count_ge20=ge,
                        report_pct_mean3=REPORT_PCT_MEAN3, report_pct_gauss3_s1=REPORT_PCT_GAUSS3_S1,
                        **{(k if k.startswith("table_") else "resid_" + k): v for k, v in resid.items()})
    n = f1.size
    print(f"{OUT}: {os.path.getsize(OUT)} bytes; count {gt} (>=: {ge}); residual mean3 {resid['mean3']} "
          f"= {100.0 * resid['mean3'] / n:.2f} % (report {REPORT_PCT_MEAN3}), gauss3 sigma 1 "
          f"{resid['gauss3_s1']} = {100.0 * resid['gauss3_s1'] / n:.2f} % (report {REPORT_PCT_GAUSS3_S1}); "
          f"int-accumulator statement {resid['intacc_mean3']} / {resid['intacc_gauss3_s1']}")
    return


if __name__ == "__main__":
    main()
