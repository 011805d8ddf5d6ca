"""The lane-group iLQR kernel (csrc/ilqr_lane_kernels.h: ilqr_group_solve_kernel) must not be launched from an instantiation that spills vector
registers (round 6: with its ~120 scalars kept in vector-register lanes, spills on top lose scalars -- wrong status words, a hang, a memory fault
were measured on a user env whose functions needed more registers).  User-env libraries ask the runtime at launch (user_env_kernels.hip.in); the
product library's own instantiations (Navigation, NavigationLQR) are held to it here, on the device assembly, without a GPU."""
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_ring_waits  # noqa: E402


@pytest.mark.skipif(check_ring_waits.hipcc_path() is None, reason="needs the device compiler (hipcc) to produce the assembly")
def test_the_librarys_lane_group_kernels_have_no_private_segment():
    src = os.path.join(ROOT, "tf-mpc_amd", "csrc", "ilqr_lane.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "ilqr_lane.s")
        subprocess.run([check_ring_waits.hipcc_path(), *check_ring_waits.FLAGS, "--cuda-device-only", "-S", src, "-o", out], check=True,
                       capture_output=True)
        text = open(out).read()
    found = re.findall(r"\.name:\s+(\S*ilqr_group_solve_kernel\S*)\n\s+\.private_segment_fixed_size:\s+(\d+)(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)",
                       text)
    assert len(found) >= 4, found          # Navigation and NavigationLQR, one and four instances per wave
    for name, private, vgprs, spills in found:
        # (beyond 256 registers the compiler parks vector registers in accumulation registers: a spill by another name)
        assert int(private) == 0 and int(spills) == 0 and int(vgprs) <= 256, (name, private, vgprs, spills)
