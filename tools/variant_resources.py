"""Register / scratch usage of each tile variant of a fused kernel compiled ALONE (diagnostic).
usage: python tools/variant_resources.py [k_step_fused|k_steps_resident] [variant indices...]   (extra -D flags: WV_DEFS)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "waves.jl_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "variants")
VARIANTS = [("AUX_NONE", "0", "RF"), ("AUX_NONE", "F_SRC", "RF"), ("AUX_NONE", "F_CYL | F_SRC", "RF"),
            ("AUX_PX", "F_EL | F_SRC", "RB"), ("AUX_PX", "F_ER | F_SRC", "RB"), ("AUX_PX", "F_ALL", "RB"),
            ("AUX_PY", "0", "RB"), ("AUX_PY", "F_ET | F_SRC", "RB"), ("AUX_PY", "F_EB | F_SRC", "RB"), ("AUX_PY", "F_ALL", "RB"),
            ("AUX_ALL", "F_EDGE | F_SRC", "RP"), ("AUX_ALL", "F_ALL", "RP"),
            ("AUX_PX", "F_EL", "RB"), ("AUX_PX", "F_ER", "RB"), ("AUX_PY", "F_ET", "RB"), ("AUX_PY", "F_EB", "RB"),
            ("AUX_ALL", "F_EDGE", "RP")]


def main():
    kern = sys.argv[1] if len(sys.argv) > 1 else "k_steps_resident"
    sel = [int(v) for v in sys.argv[2:]] or list(range(len(VARIANTS)))
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CS, "kernels_fused.hip")).read()
    k = src.index("void %s(%s)" % (kern, "JobArgs a_" if kern == "k_steps_resident" else "FusedParams p_"))
    a = src.index("if (t.aux == AUX_NONE) {", k)
    b = src.index("#undef RUN", a)
    procs = []
    for i in sel:
        A, F, R = VARIANTS[i]
        path = os.path.join(OUT, "v%d.hip" % i)
        open(path, "w").write(src[:a] + "    RUN(%s, %s, %s); (void)fl; (void)fe; (void)cyl;\n" % (A, F, R) + src[b:])
        cmd = ("hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -I%s -I%s/include "
               "%s -c %s -o %s.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2> %s.txt" % (CS, ROOT, os.environ.get("WV_DEFS", ""), path, path, path))
        procs.append(subprocess.Popen(cmd, shell=True))
        if len(procs) >= 6:
            procs.pop(0).wait()
    for p in procs:
        p.wait()
    mangled = "%d%sILi8ELi4ELi3ELi2E" % (len(kern), kern)
    for i in sel:
        t = open(os.path.join(OUT, "v%d.hip.txt" % i)).read()
        seg = t[t.index(mangled):][:2500]
        g = lambda key: re.search(key + r": (\d+)", seg).group(1)
        print("%-34s SGPR %3s VGPR %3s scratch %3s B  sgpr-spill %3s vgpr-spill %3s" % (
            "<%s, %s>" % (VARIANTS[i][0], VARIANTS[i][1]), g("TotalSGPRs"), g("VGPRs"), g(r"ScratchSize \[bytes/lane\]"),
            g("SGPRs Spill"), g("VGPRs Spill")))


if __name__ == "__main__":
    main()
