"""Instruction mix of the loops of ONE kernel in a device assembly listing (hipcc -S --cuda-device-only).
python tools/probes/asm_loops.py file.s <kernel-name-substring> [min_instructions]
For every backward branch (a loop) prints the body's instruction count by class: VALU (packed / transcendental apart),
MFMA, SALU, LDS, VMEM, waits.  Used to budget the per-step instruction counts quoted in DESIGN.md."""
import re, sys, collections

def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"): return "mfma"
    if op.startswith("v_pk_"): return "valu_pk"
    if op in ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32") or op.startswith("v_div_") or op.startswith("v_permlane"): return "valu_slow"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"): return "valu_lane"
    if op.startswith("v_accvgpr"): return "acc"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    return "other"

def main():
    path, name = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_][\w$.]*:", l) and name in l.split(":")[0]:
            start = i; break
    assert start is not None, "kernel not found"
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels, instrs = {}, []
    for l in body:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m: labels[m.group(1)] = len(instrs)
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m: labels[m.group(1)] = len(instrs); continue
        if s.endswith(":"): continue
        instrs.append(s.split(";")[0].strip())
    print("kernel instructions:", len(instrs))
    for idx, ins in enumerate(instrs):
        parts = ins.split()
        if parts[0].startswith("s_cbranch") or parts[0] == "s_branch":
            tgt = parts[-1]
            if tgt in labels and labels[tgt] <= idx:
                lo = labels[tgt]
                n = idx - lo + 1
                if n < min_n: continue
                c = collections.Counter(classify(x.split()[0]) for x in instrs[lo:idx + 1])
                ops = collections.Counter(x.split()[0] for x in instrs[lo:idx + 1])
                print(f"loop {tgt} [{lo}..{idx}] n={n}: " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
                if "-v" in sys.argv:
                    print("   ", " ".join(f"{k}:{v}" for k, v in ops.most_common(40)))
main()
