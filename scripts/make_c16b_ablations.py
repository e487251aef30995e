#!/usr/bin/env python3
"""Writes scratch/c16b/*.h: tz_conv_kernels.hip.h with ONE section of k_conv16b taken out (wrong results by design, timing
only) for scripts/gpu_c16b_ab.sh.  Result of round 5: profiles/r05/conv16b_ablations.txt."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "tezip_amd", "csrc", "tz_conv_kernels.hip.h")).read()
i0 = src.index("__global__ __launch_bounds__(NTHR, 6) void k_conv16b(const ConvArgs a) {")
i1 = src.index("// Level-0 prediction Ahat_0 = min(relu(conv3x3(r_0)), 1)")
body = src[i0:i1]
out = os.path.join(ROOT, "scratch", "c16b")
os.makedirs(out, exist_ok=True)


def variant(name, *pairs):
    b = body
    for old, new in pairs:
        assert old in b, (name, old[:50])
        b = b.replace(old, new)
    open(os.path.join(out, name + ".h"), "w").write(src[:i0] + b + src[i1:])


variant("v0")
variant("v1_noweights", ("                glds16(wsrc + (piece - P) * 256 + ln * 4, dst + piece * 256);", "                ;"))
variant("v3_noepi", ("    conv_epilogue<NT, EPI, MAP>(a, acc, n, cb, ty0, tx0, wv, lane, smem);\n}",
                     "    { float s_ = 0.f;\n      for (int mt = 0; mt < MT; ++mt) for (int nt = 0; nt < NT; ++nt) s_ += acc[mt][nt][0] + acc[mt][nt][1] + acc[mt][nt][2] + acc[mt][nt][3];\n"
                     "      if (s_ == 123.456f) a.out0[0] = s_; }\n}"))
variant("v4_noinit", ("                if (a.initf) {", "                if (false) {"), ("                } else if (a.init) {", "                } else if (false) {"))
variant("v5_nopatch",
        ("glds16(ok ? s.p + (long long)n * s.nstride + ((long long)yy * a.W + xx) * s.pstride + 4 * q : a.zero, dst + piece * 256);", ";"),
        ("                glds16(ok ? s.p + (long long)n * s.nstride + ((long long)ly * (a.W >> 1) + lx) * s.pstride + (b - 1) * 16 + 4 * q : a.zero,\n"
         "                       dst + piece * 256);", ";"))
print(sorted(os.listdir(out)))
