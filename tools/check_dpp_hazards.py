"""The K-row SA kernels add a row-broadcast operand with inline-asm `v_add_f32_dpp` (sa_kernels_impl.h: add_row_bcast_f32); the
compiler's hazard recogniser does not look inside inline asm, and gfx9 wants two wait states between a VALU write of a VGPR
and a DPP read of it.  This compiles the two bf16 units to ISA and checks every such add: python tools/check_dpp_hazards.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "point-cloud-reid_amd", "csrc")


def writes(ins):
    m = re.match(r"^(v_\S+|ds_read\S*|global_load\S*|buffer_load\S*)\s+(v\[\d+:\d+\]|v\d+)", ins)
    if not m:
        return set()
    d = m.group(2)
    if d.startswith("v["):
        a, b = map(int, re.findall(r"\d+", d))
        return set(range(a, b + 1))
    return {int(d[1:])}


def check(unit):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "u.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                               "-I" + CSRC, "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, unit)],
                              stderr=subprocess.DEVNULL)
        lines = [l.strip() for l in open(out) if l.strip() and not l.strip().startswith((";", "."))]
    n = bad = 0
    for i, l in enumerate(lines):
        if not l.startswith("v_add_f32_dpp"):
            continue
        n += 1
        src0 = int(re.findall(r"v(\d+)", l)[1])
        j, states = i - 1, 0
        while j >= 0 and states < 2:
            p = lines[j]
            if p.startswith("s_nop"):
                states += int(p.split()[1]) + 1
            else:
                if p.startswith("v_") and src0 in writes(p):
                    bad += 1
                    print("HAZARD in %s: %s -> %s" % (unit, p, l))
                states += 1
            j -= 1
    print("%s: %d v_add_f32_dpp, %d within two wait states of a VALU write of their DPP operand" % (unit, n, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if sum(check(u) for u in ("sa_kernels_bf3.hip", "sa_kernels_bf1.hip")) else 0)
