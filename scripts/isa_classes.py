"""Instruction mix of one kernel from the compiler's assembly listing (hipcc -S --cuda-device-only):
counts per class over the whole kernel body and over its main loop (the longest stretch between a
label and the backward branch to it).  python scripts/isa_classes.py file.s kernel_symbol"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
sym = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start + 1: end + 1]


def cls(op):
    if op.startswith("v_pk_"): return "VALU packed fp32 (v_pk_*)"
    if op.startswith(("v_permlane", "v_mov_b32_dpp")) or "dpp" in op: return "VALU cross-lane (permlane / dpp)"
    if op.startswith(("v_log", "v_exp", "v_sqrt", "v_rcp", "v_rsq", "v_sin", "v_cos")): return "VALU transcendental (quarter rate)"
    if op.startswith(("v_accvgpr", "v_mov", "v_readlane", "v_writelane", "v_readfirstlane", "v_cndmask", "v_swap")): return "VALU moves / selects / lane spills"
    if op.startswith(("v_fma", "v_mul_f", "v_add_f", "v_sub_f", "v_mac", "v_fmac", "v_max_f", "v_min_f", "v_max3", "v_med3")): return "VALU scalar fp32 arithmetic"
    if op.startswith("v_cmp"): return "VALU compares"
    if op.startswith("v_"): return "VALU integer / address / other"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vector memory"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep")): return "waits / barriers / nops"
    if op.startswith(("s_load", "s_buffer", "s_scratch")): return "scalar memory"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_call")): return "branches"
    if op.startswith("s_"): return "SALU"
    return "other"


def count(chunk):
    c = collections.Counter()
    for l in chunk:
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        c[cls(l.split()[0])] += 1
    return c


labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
whole = count(body)
print("kernel %s: %d instructions" % (sym, sum(whole.values())))
for k, v in whole.most_common():
    print("  %-42s %6d" % (k, v))
if loops:
    n, a, b = max(loops)
    inner = count(body[a: b + 1])
    tot = sum(inner.values())
    valu = sum(v for k, v in inner.items() if k.startswith("VALU"))
    print("main loop (%d instructions, %d of them VALU):" % (tot, valu))
    for k, v in inner.most_common():
        print("  %-42s %6d  %5.1f%%" % (k, v, 100.0 * v / tot))
