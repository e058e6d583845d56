"""Instruction census of the pair kernel from the compiler's assembly (development aid).

usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only sph_amd.hip -o sph.s
       python tools/isa_census.py sph.s
Prints, for k_pair<3,xsph,fused,narrow>: the whole kernel, and one round of the pair loop (the
loop is unrolled four times; a round runs from its LDS exchange to the end of its arithmetic)."""
import collections
import re
import sys


def classify(op):
    if op in ("v_rsq_f64_e32", "v_rcp_f64_e32", "v_sqrt_f64_e32"):
        return "FP64 transcendental"
    if op.endswith("_f64") or "_f64_" in op:
        return "FP64 arithmetic/compare/convert"
    if op.startswith("v_mov_b32_dpp"):
        return "cross-lane move (DPP)"
    if op.startswith("v_"):
        return "other vector ALU"
    if op.startswith("global_load_lds"):
        return "VMEM global->LDS copy"
    if op.startswith("global_load"):
        return "VMEM load"
    if op.startswith("global_store") or op.startswith("global_atomic"):
        return "VMEM store/atomic"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "scalar"
    return "other"


def census(lines):
    c = collections.Counter()
    for l in lines:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if m and not l.strip().startswith((";", ".")):
            c[classify(m.group(1))] += 1
    return c


text = open(sys.argv[1]).read().splitlines()
start = next(i for i, l in enumerate(text) if l.startswith("_Z6k_pairILi3ELb1ELb1ELb0EE"))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
body = text[start:end + 1]
print("k_pair<3, xsph, fused, 32-bit offsets>: %d lines of assembly" % len(body))
for k, v in sorted(census(body).items(), key=lambda kv: -kv[1]):
    print("  whole kernel  %-36s %5d" % (k, v))
# one round: from the first of a round's four global->LDS copies to the first of the next round's
# (the rounds of a super-round are unrolled; the second and third copy groups bracket a plain round)
copies = [i for i, l in enumerate(body) if "global_load_lds_dwordx4" in l]
groups = [i for k, i in enumerate(copies) if k == 0 or i - copies[k - 1] > 8]
if len(groups) >= 3:
    rnd = body[groups[1]:groups[2]]
    print("one round of the pair loop (%d lines):" % len(rnd))
    for k, v in sorted(census(rnd).items(), key=lambda kv: -kv[1]):
        print("  per round     %-36s %5d" % (k, v))
