"""VGPR liveness over the control-flow graph of one kernel's ISA listing (diagnostic; exec masks are ignored, so the
numbers are an upper bound of what the allocator sees).  Prints the pressure at every barrier / scratch access and the
registers live across the whole of a chosen line range.
usage: python tools/isa_liveness.py kernel.s [report_first_line report_last_line]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
reg_re = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = []
    for m in reg_re.finditer(tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


# ---- instructions and blocks
blocks = []  # each: dict(label, ins=[(line, op, defs, uses, text)], succ=[labels], fall=bool)
cur = {"label": "entry", "ins": [], "succ": [], "fall": True}
for n, raw in enumerate(lines):
    t = raw.split(";")[0].strip()
    if not t or t.startswith(".") and not t.endswith(":"):
        continue
    if t.endswith(":"):
        if cur["ins"] or cur["label"] == "entry":
            blocks.append(cur)
        cur = {"label": t[:-1], "ins": [], "succ": [], "fall": True}
        continue
    parts = t.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    if op in ("s_branch",) or op.startswith("s_cbranch"):
        cur["ins"].append((n + 1, op, [], [], t))
        cur["succ"].append(ops[0])
        nxt = {"label": "_after_%d" % (n + 1), "ins": [], "succ": [], "fall": True}
        cur["fall"] = op != "s_branch"
        blocks.append(cur)
        cur = nxt
        continue
    if op in ("s_endpgm", "s_setpc_b64"):
        cur["ins"].append((n + 1, op, [], [], t))
        cur["fall"] = False
        blocks.append(cur)
        cur = {"label": "_after_%d" % (n + 1), "ins": [], "succ": [], "fall": True}
        continue
    nodef = op.startswith(("s_", "global_store", "flat_store", "scratch_store", "ds_write", "ds_store", "buffer_store", "v_cmp",
                           "buffer_inv", "buffer_wbl2")) and not op.startswith(("s_waitcnt",))
    if op.startswith("s_waitcnt"):
        cur["ins"].append((n + 1, op, [], [], t))
        continue
    if nodef:
        d, u = [], [r for o in ops for r in regs(o)]
    else:
        d = regs(ops[0]) if ops else []
        u = [r for o in ops[1:] for r in regs(o)]
        # partial writes under exec / dpp / accumulate forms keep the old value alive
        if "dpp" in t or op.startswith(("v_fmac", "v_mac", "v_writelane")):
            u += d
    cur["ins"].append((n + 1, op, d, u, t))
blocks.append(cur)
index = {b["label"]: i for i, b in enumerate(blocks)}
for i, b in enumerate(blocks):
    s = [index[l] for l in b["succ"] if l in index]
    if b["fall"] and i + 1 < len(blocks):
        s.append(i + 1)
    b["s"] = s
# ---- dataflow
live_in = [set() for _ in blocks]
live_out = [set() for _ in blocks]
changed = True
while changed:
    changed = False
    for i in range(len(blocks) - 1, -1, -1):
        b = blocks[i]
        out = set()
        for s in b["s"]:
            out |= live_in[s]
        live = set(out)
        for (n, op, d, u, t) in reversed(b["ins"]):
            # a def under a divergent exec mask does not kill; we cannot see exec, so only kill in blocks that are not
            # the target of an s_and_saveexec region -- approximation: always kill (lower bound inside masked regions)
            live -= set(d)
            live |= set(u)
        if live != live_in[i] or out != live_out[i]:
            live_in[i], live_out[i] = live, out
            changed = True
# ---- per-instruction pressure
press = {}
sets = {}
for i, b in enumerate(blocks):
    live = set(live_out[i])
    for (n, op, d, u, t) in reversed(b["ins"]):
        live -= set(d)
        live |= set(u)
        press[n] = len(live)
        sets[n] = set(live)
allp = sorted(press.items())
mx = max(allp, key=lambda x: x[1])
print("max live VGPRs %d at line %d" % (mx[1], mx[0]))
for i, b in enumerate(blocks):
    for (n, op, d, u, t) in b["ins"]:
        if op in ("s_barrier",) or op.startswith("scratch_") or op == "s_sleep":
            print("%5d %-22s live %3d" % (n, op, press[n]))
if len(sys.argv) > 3:
    lo, hi = int(sys.argv[2]), int(sys.argv[3])
    common = None
    for n, s in sets.items():
        if lo <= n <= hi:
            common = set(s) if common is None else common & s
    print("live across all of %d..%d: %d registers: %s" % (lo, hi, len(common or []), sorted(common or [])))
if len(sys.argv) > 4:
    at = int(sys.argv[4])
    s = sets[at]
    print("live at %d (%d):" % (at, len(s)))
    flat = []
    for b in blocks:
        flat.extend(b["ins"])
    flat.sort()
    for r in sorted(s):
        dline = max([n for (n, op, d, u, t) in flat if r in d and n < at] or [0])
        uline = min([n for (n, op, d, u, t) in flat if r in u and n >= at] or [0])
        dt = [t for (n, op, d, u, t) in flat if n == dline]
        ut = [t for (n, op, d, u, t) in flat if n == uline]
        print("  v%-3d def %5d %-60s next use %5d %s" % (r, dline, (dt or [""])[0][:60], uline, (ut or [""])[0][:60]))
