"""checks a hipcc -S listing for the hazard the compiler cannot see: an instruction that reads or writes a VGPR which is
the destination of an inline-asm global_load still in flight (python tools/micro/asm_load_scan.py file.s kernel_substring)"""
import re, sys
src = open(sys.argv[1]).read()
sub = sys.argv[2]


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


bad = 0
for name in re.findall(r"^(\w*%s\w*):" % sub, src, flags=re.M):
    body = src[src.index(name + ":"):]
    body = body[:body.index("s_endpgm")]
    recent, n = [], 0
    for i, l in enumerate(body.split("\n")):
        t = l.strip()
        if t.startswith("global_load_dwordx4"):
            ops = [o.strip() for o in t[len("global_load_dwordx4"):].split(",")]
            recent.append(regs(ops[0]))
            continue
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
        if m:
            keep = min(int(m.group(1)), 8)          # (DMA pieces share the counter: conservative)
            recent = recent[-keep:] if keep > 0 else []
            continue
        if not t or t[0] in ";.":
            continue
        used = set()
        for k in re.findall(r"v\[\d+:\d+\]|v\d+", t):
            used |= regs(k)
        pend = set().union(*recent) if recent else set()
        if used & pend:
            print(name[-40:], i, t[:90], sorted(used & pend)[:4])
            n += 1
    print(name[-44:], "hazards:", n)
    bad += n
sys.exit(1 if bad else 0)
