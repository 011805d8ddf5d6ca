"""Static check of the hand-counted `s_waitcnt vmcnt(N)` of the LDS-DMA rings in tf-mpc_amd/csrc/ilqr_adjoint_mfma.hip.

The rings wait for "the loads of this step" by counting what is younger: vmcnt((depth - 1) * (DMA loads per step [+ stores per step])).
That is only right if the compiler emits exactly the vector-memory instructions the source counts -- with FEWER per step the wait would
return before the DMA has landed and the step would read a stale slot, silently.  This script compiles the translation unit(s) to device
assembly and checks every innermost loop that holds `global_load_lds_*` and a counted wait:

    vmcnt(N) with  N == k * (DMA loads in the loop body)                                   (a wait that counts loads only), or
                   k * (DMA loads) < N <= k * (ALL vector-memory instructions in the loop body)   (one that counts the stores too),
    k = ring depth - 1 (depth 3 for the two-tile kernels, 4 otherwise)

    python tools/check_ring_waits.py [part ...]        parts 0..9 (default: all); exit status 1 on a violation
Importable: check_part(part) -> list of findings (tests/test_ring_waits_cpu.py runs the two cfg5 parts)."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tf-mpc_amd", "csrc", "ilqr_adjoint_mfma.hip")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form".split()


def assembly(part):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "x.s")
        subprocess.run([hipcc_path(), *FLAGS, f"-DTFMPC_AM_PART={part}", "--cuda-device-only", "-S", SRC, "-o", out],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read()


def kernels(text):
    """name -> list of (kind, payload): ('i', instruction) | ('l', label)"""
    out, cur = {}, None
    for line in text.split("\n"):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m and "ilqr_adjoint_mfma_kernel" in m.group(1):
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None or not s or s.startswith(";"):
            continue
        if s.startswith("s_endpgm"):
            cur = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur.append(("l", m.group(1)))
        elif not s.startswith(".") and not s.endswith(":"):
            cur.append(("i", s.split(";")[0].strip()))
    return out


def check_kernel(name, items):
    findings, labels, instrs = [], {}, []
    for kind, v in items:
        if kind == "l":
            labels[v] = len(instrs)
        else:
            instrs.append(v)
    loops = []
    for idx, ins in enumerate(instrs):
        op = ins.split()[0]
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = ins.split()[-1]
            if tgt in labels and labels[tgt] <= idx:
                loops.append((labels[tgt], idx))
    # round 6: the DMA helpers overwrite M0 without saving it -- sound only while nothing else in the kernel uses M0
    for idx, ins in enumerate(instrs):
        if re.search(r"\bm0\b", ins) and not (re.match(r"^s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi)$", ins) and idx + 2 < len(instrs) and instrs[idx + 1].startswith("s_nop")
                                               and instrs[idx + 2].startswith("global_load_lds")):
            findings.append(f"{name[:60]}: `{ins}` uses M0 outside a DMA issue (the DMA helpers do not preserve it)")
    checked = 0
    for lo, hi in loops:
        if any(l2 != (lo, hi) and lo <= l2[0] and l2[1] <= hi for l2 in loops):
            continue                                           # not innermost
        body = instrs[lo:hi + 1]
        dma = sum(1 for x in body if x.startswith("global_load_lds"))
        if dma == 0:
            continue
        vm = sum(1 for x in body if re.match(r"^(global_|buffer_|flat_|scratch_)", x))
        all_waits = [int(m.group(1)) for x in body for m in [re.search(r"vmcnt\((\d+)\)", x)] if m and x.startswith("s_waitcnt") and int(m.group(1)) > 0]
        waits = sorted(set(all_waits))
        if not waits:
            continue
        # the compiler may put several time steps into one trip of the loop (each with its own waits): per-step counts
        steps = max(1, len(all_waits) // len(waits))
        if dma % steps or vm % steps:
            findings.append(f"{name[:70]}: loop [{lo}..{hi}]: {len(all_waits)} counted waits, {dma} DMA loads, {vm} vector-memory instructions: "
                            "not a whole number per time step")
            continue
        dma, vm = dma // steps, vm // steps
        checked += 1
        # ring depth from the kernel's template arguments (kRingDepth = NT == 2 ? 3 : 4): ..._kernelILi<KIND>ELi<NT>E...
        m = re.search(r"ilqr_adjoint_mfma_kernelILi\d+ELi(\d+)E", name)
        k = (3 if m and m.group(1) == "2" else 4) - 1
        for n in waits:
            if n == k * dma:
                continue                                       # "the loads of the younger steps": exactly what the compiler emitted
            if k * dma < n <= k * vm:
                continue                                       # loads + stores of the younger steps: no more than are really issued
            if n * (k + 1) == k * dma and steps == 1:
                continue                                       # a ring's PROLOGUE with its first, peeled step (no time loop of its own left around it): depth
                                                               # steps' loads issued, the wait leaves the depth - 1 younger steps' in flight
            findings.append(f"{name[:70]}: loop [{lo}..{hi}]: vmcnt({n}) with ring depth {k + 1}, {dma} DMA loads and {vm - dma} other "
                            f"vector-memory instructions per step: " + ("the wait counts MORE operations than a step issues -- it would return early"
                                                                         if n > k * vm else "the wait is not a whole number of steps' loads"))
    return findings, checked


def hipcc_path():
    """The device compiler: PATH, then $ROCM_PATH/bin, then /opt/rocm/bin; None if there is none (a CPU-only box without ROCm)."""
    import shutil
    for cand in (shutil.which("hipcc"), os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def check_part(part):
    findings, checked = [], 0
    for name, items in kernels(assembly(part)).items():
        f, c = check_kernel(name, items)
        findings += f
        checked += c
    return findings, checked


if __name__ == "__main__":
    parts = [int(p) for p in sys.argv[1:]] or list(range(10))
    bad = 0
    for p in parts:
        findings, checked = check_part(p)
        print(f"part {p}: {checked} ring loops checked, {len(findings)} violations")
        for f in findings:
            print("  ", f)
        bad += len(findings)
    sys.exit(1 if bad else 0)
