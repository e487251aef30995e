"""TEZIP command line for the MI355X build.

Behaviourally equivalent to the reference CLI (/root/reference/src/tezip.py:10-100): same
flags, same validation order and the same messages on stdout, process exit code 0 on
validation errors.  The structure is this build's own: flags come from a table, validation is a
list of (predicate, message key) rules evaluated in the reference's order, and the device probe
asks the HIP library instead of TensorFlow.  `-l` trains with tezip_amd/train.py (PyTorch
autograd) on .npy stacks written by tezip_amd/train_data_create.py.
"""
import argparse
import os
import sys

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    __package__ = "tezip_amd"

from . import compress, decompress, train  # noqa: E402

# (short, long, argparse keywords) -- tezip.py:88-99
FLAG_TABLE = (
    ("-l", "--learn", dict(type=str, nargs=2, metavar=("model", "dir"), dest="learn")),
    ("-c", "--compress", dict(type=str, nargs=3, metavar=("model", "dir", "file"), dest="compress")),
    ("-u", "--uncompress", dict(type=str, nargs=3, metavar=("model", "file", "dir"), dest="uncompress")),
    ("-p", "--preprocess", dict(type=int, nargs=1, metavar="warm_up_num", dest="preprocess")),
    ("-w", "--window", dict(type=int, nargs=1, metavar="window_size", dest="window")),
    ("-t", "--threshold", dict(type=float, nargs=1, metavar="MSE_threshold", dest="threshold")),
    ("-m", "--mode", dict(type=str, nargs=1, metavar="mode", dest="mode")),
    ("-b", "--bound", dict(type=float, nargs="*", metavar="value", dest="bound", default=None)),
    ("-f", "--force", dict(action="store_true")),
    ("-v", "--verbose", dict(action="store_true")),
    ("-n", "--no_entropy", dict(action="store_false")),  # store_false: entropy remap is on by default
    # not in the reference: compress once per candidate -w and keep the smallest output (BASELINE.json configs[4];
    # tezip_amd/sweep.py).  No value = 5 10 15 20 25 30 35 40.  Takes the place of -w / -t.
    # not in the reference: byte-shuffled payload (flagged in the trailer; the reference cannot read such a file)
    (None, "--shuffle", dict(action="store_true")),
    (None, "--sweep", dict(type=int, nargs="*", metavar="window_size", dest="sweep", default=None)),
    # not in the reference: the predictor's arithmetic contract.  0 / absent = by frame size (TZ-PA2 from
    # 256x256 pixels on), 1 = TZ-PA1 (what files compressed by builds before round 4 need for -u), 2 = TZ-PA2.  The same
    # value must be given to -c and -u: the reference's file format has no field for it.
    (None, "--pa", dict(type=int, choices=(0, 1, 2), default=None, dest="pa")),
)

TEXT = {
    "several": ("Please select only one of learn or compress or uncompress.",
                "Command to check the options is -h or --help"),
    "nothing": ("Please mode select!", "learn or compress or uncompress.",
                "Command to check the options is -h or --help"),
    "no_p": ("Please specify the -p or --preprocess option!", "warm up num."),
    "no_window": ("Please specify the window size(-w or --window) or MSE threshold(-t or --threshold) option!",
                  "Select window size for SWP and MSE threshold for DWP."),
    "two_windows": ("Please select only one of window size(-w or --window) or MSE threshold(-t or --threshold)!",
                    "Select window size for SWP and MSE threshold for DWP."),
    "bad_mode": ("Please specify the -m or --mode correctly!", "'abs' or 'rel' or 'absrel' or 'pwrel'."),
    "no_bound": ("Please specify the -b or --bound option!", "error bound value."),
    "bound_count": ("If the -m or --mode is 'abs' or 'rel' or 'pwrel', enter one for -b or --bound. : value",
                    "If the -m or --mode is 'absrel', enter two in -b or --bound. : abs_value rel_value"),
}
BOUNDS_WANTED = {"abs": 1, "rel": 1, "pwrel": 1, "absrel": 2}


def build_parser():
    parser = argparse.ArgumentParser(prog="TEZIP", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for short, long_, kw in FLAG_TABLE:
        parser.add_argument(*([short, long_] if short else [long_]), **kw)
    return parser


def probe_gpu(force_cpu):
    """tezip.py:12-21 asked TensorFlow for a GPU; here a context on device 0 must open."""
    if force_cpu:
        return False
    try:
        from . import _lib
        _lib.Context(0).close()
        return True
    except Exception:
        return False


def complain(key):
    print("ERROR")
    for line in TEXT[key]:
        print(line)


def check_compress(arg):
    """Reference order (tezip.py:40-84).  Returns None when the request is complete, else the key
    of the message to print; prints the mode name where the reference does."""
    if arg.preprocess is None:
        return "no_p"
    have_sweep = getattr(arg, "sweep", None) is not None
    have_w, have_t = arg.window is not None or have_sweep, arg.threshold is not None
    if not have_w and not have_t:
        return "no_window"
    if (have_w and have_t) or (have_sweep and arg.window is not None):  # --sweep takes the place of -w / -t: not beside them
        return "two_windows"
    if arg.mode is None:  # the reference raises a TypeError here; report it as a bad mode instead
        return "bad_mode"
    print(arg.mode[0])
    if arg.mode[0] not in BOUNDS_WANTED:
        return "bad_mode"
    if not arg.bound:
        return "no_bound"
    if len(arg.bound) != BOUNDS_WANTED[arg.mode[0]]:
        return "bound_count"
    return None


def _main(arg):
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:  # launched by torch.distributed.run: one rank per GPU
        from . import dist as tzdist
        tzdist.init_from_env()
    gpu = probe_gpu(arg.force)
    print("GPU MODE" if gpu else "CPU MODE")
    chosen = [name for name in ("learn", "compress", "uncompress") if getattr(arg, name) is not None]
    if len(chosen) > 1:
        return complain("several")
    if not chosen:
        return complain("nothing")
    if chosen[0] == "learn":
        print("train mode")
        return train.run(arg.learn[0], arg.learn[1], arg.verbose)
    if chosen[0] == "uncompress":
        print("uncompress mode")
        model, src, dst = arg.uncompress
        return decompress.run(model, src, dst, gpu, arg.verbose)
    print("compress mode")
    problem = check_compress(arg)
    if problem:
        return complain(problem)
    model, src, dst = arg.compress
    if arg.sweep is not None and arg.window is None:
        from . import sweep
        return sweep.run(model, src, dst, arg.preprocess[0], arg.sweep or None, arg.mode[0], arg.bound, arg.verbose,
                         arg.no_entropy)
    window = arg.window[0] if arg.window is not None else None
    threshold = arg.threshold[0] if arg.threshold is not None else None
    return compress.run(model, src, dst, arg.preprocess[0], window, threshold, arg.mode[0], arg.bound, gpu,
                        arg.verbose, arg.no_entropy, SHUFFLE=arg.shuffle)


def main(arg):
    """--pa travels to the library as TEZIP_PA (every context made during this run starts with it: tz_ctx_create); the
    variable is put back when the run ends, so that a caller that drives main() in-process is left as it was."""
    if getattr(arg, "pa", None) is None:
        return _main(arg)
    before = os.environ.get("TEZIP_PA")
    os.environ["TEZIP_PA"] = str(arg.pa)
    try:
        return _main(arg)
    finally:
        if before is None:
            os.environ.pop("TEZIP_PA", None)
        else:
            os.environ["TEZIP_PA"] = before


if __name__ == "__main__":
    main(build_parser().parse_args())
