import re, sys, collections
# for every kernel in a listing: the smallest loops that hold LDS-DMA loads, and every s_waitcnt with a vmcnt inside them
path = sys.argv[1]
txt = open(path).read().split("\n")
starts = [i for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l)]
for st in starts:
    name = txt[st].split(":")[0]
    t = re.search(r'ILi(\d+)ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELi(\d)E', name)
    end = next(i for i in range(st, len(txt)) if txt[i].strip().startswith("s_endpgm"))
    labels, instrs = {}, []
    for l in txt[st:end + 1]:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m: labels[m.group(1)] = len(instrs); continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"): continue
        instrs.append(s.split(";")[0].strip())
    loops = []
    for idx, ins in enumerate(instrs):
        parts = ins.split()
        if parts[0].startswith("s_cbranch") or parts[0] == "s_branch":
            tgt = parts[-1]
            if tgt in labels and labels[tgt] <= idx: loops.append((labels[tgt], idx))
    occ = [i for i, x in enumerate(instrs) if "global_load_lds" in x]
    seen = set()
    for o in occ:
        enc = [(b - a + 1, a, b) for a, b in loops if a <= o <= b]
        if not enc: continue
        n, a, b = min(enc)
        if (a, b) in seen: continue
        seen.add((a, b))
        w = [x for x in instrs[a:b + 1] if x.startswith("s_waitcnt") and "vmcnt" in x]
        c = collections.Counter(re.search(r"vmcnt\((\d+)\)", x).group(1) for x in w)
        flag = "  <== vmcnt(0) inside" if "0" in c else ""
        print(f"KIND {t.group(1)} NT {t.group(2)} VW {t.group(3)} PK {t.group(4)} BF {t.group(5)} NW {t.group(6)}: loop [{a}..{b}] n={n} dma={sum(1 for i in occ if a <= i <= b)} vmcnt waits {dict(c)}{flag}")
