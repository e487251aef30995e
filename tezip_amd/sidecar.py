"""tezip_amd.json -- what the reference's three files cannot say about a compressed directory.

A lossless decode needs the decoder's predictions bit-identical to the encoder's
(/root/reference/src/decompress.py:252-253: `X_hat * 255 - difference`).  The reference leaves the
predictor's float32 arithmetic to Keras/TensorFlow/cuDNN; this build fixes it as an arithmetic
CONTRACT (TZ-PA1 direct chains / TZ-PA2 Winograd chains, include/tezip_hip.h tz_set_contract), and
which one encoded a stream is not derivable from the stream.  The reference's decoder opens exactly
filename.txt, key_frame.dat and entropy.dat (decompress.py:48-103), so a fourth file is invisible to
it; `-c` writes this one, `-u` adopts it:

  {"format": 1, "arithmetic_contract": "TZ-PA2", "contract": 2, "tz_version": 101,
   "arch": "gfx950", "padded_frame": [512, 512], "weights_sha256": "...", "stack": [80, 512, 512, 0]}

`stack` = [frames, height, width, warm_up] (round 6, optional): the reference stores these in the LAST seven
values of entropy.dat (compress.py:390-394), so a decoder learns it only when the whole payload is decompressed; with
it here the decoder runs its rollout WHILE entropy.dat is being decompressed (decompress._run_streaming) and checks the
trailer against it afterwards.

  * sidecar present: the decoder runs under its contract; a --pa / TEZIP_PA that contradicts it is an
    error (the output would be off by one grey level on a fraction of 'lossless' samples, silently);
    a model whose weights hash differs is an error too (the output would be noise);
  * sidecar absent (a directory written by the reference, or by a build before round 5): the rule of
    rounds 1-4 -- --pa / TEZIP_PA if given, else by padded frame size.
"""
import hashlib
import json
import os

NAME = "tezip_amd.json"
FORMAT = 1


class SidecarMismatch(ValueError):
    pass


def weights_sha256(wts):
    h = hashlib.sha256()
    for w in wts:
        h.update(memoryview(w if w.flags["C_CONTIGUOUS"] else w.copy()).cast("B"))
    return h.hexdigest()


def requested_contract():
    """What the command line (--pa, exported as TEZIP_PA by tezip.py) asked for: 1, 2 or None."""
    v = os.environ.get("TEZIP_PA", "").strip()
    return int(v) if v in ("1", "2") else None


def write(out_dir, contract, wts, hp, wp, stack=None):
    from . import _lib
    if contract not in (1, 2):
        raise ValueError("contract must be 1 or 2, not %r" % (contract,))
    doc = {"format": FORMAT, "arithmetic_contract": "TZ-PA%d" % contract, "contract": int(contract),
           "tz_version": int(_lib.load().tz_version()), "arch": "gfx950", "padded_frame": [int(hp), int(wp)],
           "weights_sha256": weights_sha256(wts)}
    if stack is not None:
        doc["stack"] = [int(v) for v in stack]
    with open(os.path.join(out_dir, NAME), "w", encoding="UTF-8") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    return doc


def read(data_dir):
    """The sidecar of a compressed directory, or None when there is none.  A file that is there but unreadable is an
    error, not 'absent': falling back to a guess is exactly the silent failure this file exists to prevent."""
    path = os.path.join(data_dir, NAME)
    if not os.path.exists(path):
        return None
    try:
        with open(path, "r", encoding="UTF-8") as f:
            doc = json.load(f)
        if doc.get("format") != FORMAT or doc.get("contract") not in (1, 2):
            raise ValueError("format %r, contract %r" % (doc.get("format"), doc.get("contract")))
    except (OSError, ValueError) as e:
        raise SidecarMismatch("%s is damaged (%s); remove it only if you know which --pa the directory was compressed "
                              "with" % (path, e))
    return doc


def stack_of(doc):
    """(nt, H, W, warm_up) when the sidecar records a plausible stack, else None (older sidecars, the reference's directories)."""
    st = doc.get("stack") if doc else None
    if (isinstance(st, list) and len(st) == 4 and all(isinstance(v, int) for v in st) and all(1 <= v <= 32767 for v in st[:3])
            and 0 <= st[3] < st[0]):
        return tuple(st)
    return None


def resolve(doc, wts=None):
    """The contract a decoder must run under given the sidecar `doc` (may be None) and the command line.  Returns 1, 2 or
    None (= no sidecar and nothing asked for: by frame size).  Raises SidecarMismatch on a contradiction."""
    asked = requested_contract()
    if doc is None:
        return asked
    if asked is not None and asked != doc["contract"]:
        raise SidecarMismatch("this directory was compressed under %s (%s), --pa / TEZIP_PA asks for TZ-PA%d: a decode under "
                              "another arithmetic contract is silently off by one grey level on part of the samples.  Drop "
                              "--pa (the recorded contract is adopted)." % (doc["arithmetic_contract"], NAME, asked))
    if wts is not None and doc.get("weights_sha256") and doc["weights_sha256"] != weights_sha256(wts):
        raise SidecarMismatch("the model given to -u is not the model this directory was compressed with (%s records "
                              "another weights_sha256): the decoder's predictions would not be the encoder's." % NAME)
    return doc["contract"]
