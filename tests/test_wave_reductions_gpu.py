"""common.h's wave reductions run on v_permlane32_swap / v_permlane16_swap / DPP row_ror instead of __shfl_xor (= ds_bpermute_b32 on gfx950).
The claim that every lane ends with the SAME BITS as the __shfl_xor butterfly (so no kernel's result moved) is checked on the device by
tools/wave_probe.py: wave_sum, wave_max, the 16- / 32- / 8-lane and cross-group butterflies of the decode attention kernels and the
(value, index) argmax butterfly, 65 536 waves x 64 lanes per input class (normal, wide exponents, signed zeros / inf / denormals, cancelling)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_wave_reductions_give_the_bits_of_the_shuffle_butterflies():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wave_probe.py")], capture_output=True, text=True, timeout=600, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert out.count("bit-identical") == 4 and "DIFFERENT" not in out, out[-3000:]
