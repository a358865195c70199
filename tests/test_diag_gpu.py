"""The diagnostics of csrc/diag.hip (bench.py's `board` object): they are not on any data path; what is checked is that
they run on the core's device and stream, return plausible figures for an MI355X and refuse nonsense."""
import pytest

from cudavideostream_amd import lib
from cudavideostream_amd.core import CUDACore

pytestmark = pytest.mark.gpu


def test_probes_return_plausible_figures():
    with CUDACore(64, 48) as core:
        mhz = core.probe_clock(50)
        rd = core.probe_hbm_read(256)
        wide = core.probe_hbm_write(256)
        narrow = core.probe_hbm_write(256, narrow=True)
        assert 500 < mhz < 4000
        assert 1000 < rd < 9000 and 500 < wide < 9000 and 100 < narrow < 9000
        # the narrow form (an index and a value per lane) cannot beat whole lines by much
        assert narrow < 1.5 * wide


def test_probes_refuse_nonsense():
    with CUDACore(64, 48) as core:
        for bad in (0, 8, 1 << 20):
            with pytest.raises(lib.Mi355Error):
                core.probe_hbm_read(bad)
            with pytest.raises(lib.Mi355Error):
                core.probe_hbm_write(bad)
        with pytest.raises(lib.Mi355Error):
            core.probe_clock(0)
