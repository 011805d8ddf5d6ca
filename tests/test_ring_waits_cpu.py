"""Static check of the LDS-DMA rings' hand-counted waits (not a GPU test -- it reads the device ASSEMBLY).

The cfg5 kernels (tf-mpc_amd/csrc/ilqr_adjoint_mfma.hip) prefetch their inputs with `global_load_lds_*` into a ring of LDS slots and
wait with `s_waitcnt vmcnt(N)`, N counted by hand from the number of vector-memory instructions a time step issues.  If a compiler or
flag change made a step issue FEWER of them, the wait would return before the DMA has landed and the step would silently read a stale
slot.  tools/check_ring_waits.py compiles a translation unit to assembly and holds every counted wait against what the compiler
really emitted in its loop; here it runs on the two translation units of the BASELINE configs[4] kernels (HVAC and Reservoir at two
tiles), and on a doctored listing that it must reject."""

import concurrent.futures
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_ring_waits  # noqa: E402


@pytest.mark.skipif(check_ring_waits.hipcc_path() is None, reason="needs the device compiler (hipcc) to produce the assembly")
def test_the_cfg5_kernels_wait_for_what_the_compiler_emits():
    """Parts 0, 4 and 8: HVAC, Reservoir and the Reservoir chain instantiation (round 5) at two tiles; part 9 (round 6): the one-tile chain form that
    res4 takes.  Since round 6 the same pass also holds every kernel to "M0 is written by the DMA issue and used by nothing else" (the helpers
    overwrite it without saving it)."""
    parts = (0, 4, 8, 9)
    with concurrent.futures.ThreadPoolExecutor(max_workers=2) as pool:       # (the work is in hipcc child processes)
        results = list(pool.map(check_ring_waits.check_part, parts))
    for part, (findings, checked) in zip(parts, results):
        assert checked >= (20, 20, 3, 2)[parts.index(part)], (part, checked)                                # the loops were found at all
        assert not findings, (part, findings[:3])


def test_another_use_of_m0_is_reported():
    text = "\n".join(["_ZN5tfmpc12_GLOBAL__N_124ilqr_adjoint_mfma_kernelILi4ELi2ELi4ELi1ELb0ELi1EEEv8TfmpcEnv:", "s_mov_b32 m0, s5", "s_nop 0",
                      "global_load_lds_dwordx4 v[1:2], off", "s_mov_b32 s7, m0", "s_endpgm"])
    (name, items), = check_ring_waits.kernels(text).items()
    findings, _ = check_ring_waits.check_kernel(name, items)
    assert len(findings) == 1 and "M0" in findings[0]


def _listing(n_wait, stores):
    body = ["global_load_lds_dwordx4 v[1:2], off", "global_load_lds_dwordx4 v[3:4], off", "global_load_lds_ubyte v[5:6], off"]
    body += [f"global_store_dwordx4 v[{7 + i}:{8 + i}], v[20:23], off" for i in range(stores)]
    return "\n".join(["_ZN5tfmpc12_GLOBAL__N_124ilqr_adjoint_mfma_kernelILi4ELi2ELi4ELi1ELb0ELi1EEEv8TfmpcEnv:", ".LBB0_1:"]
                     + body + [f"s_waitcnt vmcnt({n_wait})", "v_add_f32 v0, v1, v2", "s_cbranch_scc1 .LBB0_1", "s_endpgm"])


def test_a_wait_that_counts_more_than_a_step_issues_is_rejected():
    check = lambda text: [f for name, items in check_ring_waits.kernels(text).items() for f in check_ring_waits.check_kernel(name, items)[0]]
    assert check(_listing(16, 5)) == []                      # 2 x (3 loads + 5 stores): the stored rollout as written
    assert check(_listing(6, 0)) == []                       # 2 x 3 loads: a search rollout
    bad = check(_listing(16, 4))                             # the compiler merged two stores: 2 x (3 + 4) = 14 < 16
    assert len(bad) == 1 and "return early" in bad[0]
    assert len(check(_listing(5, 0))) == 1                   # not a whole number of steps' loads
