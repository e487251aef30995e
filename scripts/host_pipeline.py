#!/usr/bin/env python3
"""Host pipeline of `tezip.py -c` / `-u` on the cfg3 data (80 and 320 PNGs of 512x512): warm wall
time per stage (TEZIP_TIMING) and peak RSS of a fresh process per sequence length
(overlapped stages, host memory independent of nt)."""
import os
import resource
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(m, d, out):
    os.environ["TEZIP_TIMING"] = "1"
    from tezip_amd import compress, decompress
    compress.run(m, d, out + "_warmup", 0, 20, None, "abs", [2.0], True, False, True)   # library load, HIP start-up
    print("---- warm run", file=sys.stderr)
    t0 = time.perf_counter()
    compress.run(m, d, out, 0, 20, None, "abs", [2.0], True, False, True)
    t1 = time.perf_counter()
    os.environ.pop("TEZIP_TIMING")
    decompress.run(m, out, out + "_u", True, False)
    t2 = time.perf_counter()
    sizes = {f: os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)}
    print("RESULT nt=%s compress.run %.3f s, decompress.run %.3f s, maxrss %.0f MB, files %s" % (
        len(os.listdir(d)), t1 - t0, t2 - t1, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, sizes))


def main():
    if len(sys.argv) == 5 and sys.argv[1] == "child":
        return child(*sys.argv[2:])
    from PIL import Image
    from tezip_amd import synth, weights
    from tezip_amd.prednet import PredNetConfig
    tmp = tempfile.mkdtemp(prefix="tzhost_")
    cfg = PredNetConfig()
    m = os.path.join(tmp, "model")
    weights.save_model(m, cfg, cfg.init_weights(123), 512, 512)
    frames = synth.turbulence(320, 512, 512, seed=3)
    for nt in (80, 320):
        d = os.path.join(tmp, "data%d" % nt)
        os.mkdir(d)
        for t in range(nt):
            Image.fromarray(frames[t]).save(os.path.join(d, "f%03d.png" % t))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", m, d, os.path.join(tmp, "c%d" % nt)],
                           capture_output=True, text=True)
        print(r.stderr[-3000:])
        print(r.stdout[-1500:])


if __name__ == "__main__":
    main()
