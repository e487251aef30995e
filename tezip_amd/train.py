"""PredNet trainer: the job of the reference's train.py, on PyTorch autograd.

What is reproduced from /root/reference/src/train.py:43-112: the model (stack sizes
(3,48,96,192), 3x3 filters, Keras default initialisers), nt = 2 frames per sample, the "L_0"
loss = mean absolute value of the time-weighted ([0, 1]) layer-weighted ([1,0,0,0]) error-unit
means, Adam (lr 1e-3, 1e-4 from epoch 75; Keras defaults beta 0.9/0.999, eps 1e-7), 100 epochs of
5 samples at batch 1, 2 validation sequences, best-validation checkpoint.  Samples are nt
consecutive frames of one source, start drawn from all valid starts (data_utils.py:29-30).
The forward pass here is ordinary torch conv2d (any device torch has); it is the same function
as the inference path's TZ-PA1 arithmetic up to float32 summation order, which training does
not need bit for bit.
Data: X_train.hkl / sources_train.hkl / X_val.hkl / sources_val.hkl (hickle 4.0.1 layout as restated in
tezip_amd/hkl.py, written by tezip_amd.train_data_create) or the same four names as .npy stacks.
Output: prednet_model.json + prednet_weights.hdf5 in WEIGHTS_DIR -- the reference's two files in Keras' layout
(tezip_amd/weights.py), read by compress.run / decompress.run here.  Whether the reference's Keras 2.2.4 reads
them is parity unpinned (no Keras / h5py in this image, no fixture in the reference): see weights.py.
"""
import os

import numpy as np

from . import weights as W
from .prednet import PredNetConfig


def build_model(cfg, weight_list=None, seed=123):
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    class PredNetTorch(nn.Module):
        def __init__(self):
            super().__init__()
            L, st, rs = cfg.nb_layers, cfg.stack_sizes, cfg.R_stack_sizes
            self.L = L

            def conv(cin, cout):
                return nn.Conv2d(cin, cout, 3, padding=1)

            self.a = nn.ModuleList([conv(2 * st[l], st[l + 1]) for l in range(L - 1)])
            self.ahat = nn.ModuleList([conv(rs[l], st[l]) for l in range(L)])
            gin = [rs[l] + 2 * st[l] + (rs[l + 1] if l < L - 1 else 0) for l in range(L)]
            self.c = nn.ModuleList([conv(gin[l], rs[l]) for l in range(L)])
            self.f = nn.ModuleList([conv(gin[l], rs[l]) for l in range(L)])
            self.i = nn.ModuleList([conv(gin[l], rs[l]) for l in range(L)])
            self.o = nn.ModuleList([conv(gin[l], rs[l]) for l in range(L)])
            self.load_keras_list(weight_list if weight_list is not None else cfg.init_weights(seed))

        def _ordered(self):  # Keras weight-list order (prednet.py:212): a, ahat, c, f, i, o
            return [m for grp in (self.a, self.ahat, self.c, self.f, self.i, self.o) for m in grp]

        def load_keras_list(self, ws):
            with torch.no_grad():
                for m, k, b in zip(self._ordered(), ws[0::2], ws[1::2]):
                    m.weight.copy_(torch.from_numpy(np.ascontiguousarray(k)).permute(3, 2, 0, 1))  # HWIO -> OIHW
                    m.bias.copy_(torch.from_numpy(np.ascontiguousarray(b)))

        def keras_list(self):
            out = []
            for m in self._ordered():
                out.append(m.weight.detach().cpu().permute(2, 3, 1, 0).contiguous().numpy())
                out.append(m.bias.detach().cpu().numpy().copy())
            return out

        def forward(self, x, output="error"):
            """x: (B, T, 3, H, W) in [0,1].  output 'error': (B, T, L) layer means of the error
            units (prednet.py:297-301); 'prediction': (B, T, 3, H, W) frame predictions."""
            B, T, _, H, Wd = x.shape
            L, st, rs = self.L, cfg.stack_sizes, cfg.R_stack_sizes
            hs = lambda v: torch.clamp(0.2 * v + 0.5, 0, 1)  # Keras hard_sigmoid  # noqa: E731
            r = [x.new_zeros(B, rs[l], H >> l, Wd >> l) for l in range(L)]
            c = [t.clone() for t in r]
            e = [x.new_zeros(B, 2 * st[l], H >> l, Wd >> l) for l in range(L)]
            outs = []
            for t in range(T):
                a = x[:, t]
                rn, cn = [None] * L, [None] * L
                for l in reversed(range(L)):
                    inp = [r[l], e[l]] + ([F.interpolate(rn[l + 1], scale_factor=2, mode="nearest")] if l < L - 1 else [])
                    z = torch.cat(inp, 1)
                    cn[l] = hs(self.f[l](z)) * c[l] + hs(self.i[l](z)) * torch.tanh(self.c[l](z))
                    rn[l] = hs(self.o[l](z)) * torch.tanh(cn[l])
                errs = []
                for l in range(L):
                    ahat = torch.relu(self.ahat[l](rn[l]))
                    if l == 0:
                        ahat = torch.clamp(ahat, max=1.0)
                        pred = ahat
                    e[l] = torch.cat([torch.relu(ahat - a), torch.relu(a - ahat)], 1)
                    errs.append(e[l].flatten(1).mean(1))
                    if l < L - 1:
                        a = F.max_pool2d(torch.relu(self.a[l](e[l])), 2)
                r, c = rn, cn
                outs.append(pred if output == "prediction" else torch.stack(errs, 1))
            return torch.stack(outs, 1)

    return PredNetTorch()


def possible_starts(sources, nt, mode="all", N_seq=None):
    """data_utils.py:28-45.  'all' (what train.py uses): every start whose nt frames come from one source -- the
    reference iterates range(N - nt), so the last valid start is never offered (kept); 'unique': each frame in at most one
    sequence; N_seq keeps the first N_seq of them (the validation generator of train.py:90).  Pinned to the reference's own
    SequenceGenerator by tests/golden/ref_train.npz."""
    n = len(sources)
    if mode == "all":
        starts = [i for i in range(n - nt) if sources[i] == sources[i + nt - 1]]
    elif mode == "unique":
        starts, cur = [], 0
        while cur < n - nt + 1:
            if sources[cur] == sources[cur + nt - 1]:
                starts.append(cur)
                cur += nt
            else:
                cur += 1
    else:
        raise ValueError("sequence_start_mode must be in {all, unique}")
    if N_seq is not None and len(starts) > N_seq:
        starts = starts[:N_seq]
    return np.array(starts, dtype=np.int64)


def sample(X, start, nt):
    """One training sample (data_utils.py:58-61,70-71): nt consecutive frames as float32 / 255, channels last."""
    return np.asarray(X[start:start + nt]).astype(np.float32) / 255


def l0_loss(errors, nt):
    """train.py:56-72: layer weights [1,0,..], time weights [0, 1/(nt-1), ...], MAE against 0."""
    tw = errors.new_full((nt,), 1.0 / (nt - 1))
    tw[0] = 0.0
    return (errors[:, :, 0] * tw).sum(1).abs().mean()


def run(WEIGHTS_DIR, DATA_DIR, VERBOSE, nb_epoch=100, samples_per_epoch=5, N_seq_val=2, nt=2, seed=123,
        stack_sizes=(3, 48, 96, 192), device=None):
    import torch
    def load(name):
        """The reference's hickle files (train.py:24-29, data_utils.py:14-15) or .npy stacks."""
        h = os.path.join(DATA_DIR, name + ".hkl")
        if os.path.exists(h):
            from . import hkl
            return hkl.load(h)
        return np.load(os.path.join(DATA_DIR, name + ".npy"), mmap_mode="r" if name.startswith("X_") else None)

    try:
        X, src, Xv, srcv = load("X_train"), np.asarray(load("sources_train")), load("X_val"), np.asarray(load("sources_val"))
    except (OSError, ValueError) as e:
        print("ERROR: No such file or directory:", os.path.join(DATA_DIR, "X_train.hkl"))
        print("(build the training set with `python -m tezip_amd.train_data_create`)")
        print("\nORIGINAL ERROR MESSAGE:", e)
        exit()
    dev = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    cfg = PredNetConfig(stack_sizes=stack_sizes)
    hp, wp = X.shape[1], X.shape[2]
    model = build_model(cfg, seed=seed).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-7)
    starts = rng.permutation(possible_starts(src, nt))
    vstarts = possible_starts(srcv, nt, N_seq=N_seq_val)
    if len(starts) == 0 or len(vstarts) == 0:
        print("ERROR: not enough consecutive frames per folder for nt =", nt)
        exit()

    def batch(arr, i):
        return torch.from_numpy(sample(arr, i, nt)).to(dev).permute(0, 3, 1, 2)[None]

    os.makedirs(WEIGHTS_DIR, exist_ok=True)
    best, cursor, history = float("inf"), 0, []
    for epoch in range(nb_epoch):
        for g in opt.param_groups:
            g["lr"] = 1e-3 if epoch < 75 else 1e-4  # train.py:105
        model.train()
        tl = 0.0
        for _ in range(samples_per_epoch):
            i = int(starts[cursor % len(starts)])
            cursor += 1
            opt.zero_grad()
            loss = l0_loss(model(batch(X, i)), nt)
            loss.backward()
            opt.step()
            tl += float(loss.detach())
        model.eval()
        with torch.no_grad():
            vl = float(np.mean([float(l0_loss(model(batch(Xv, int(i))), nt)) for i in vstarts]))
        history.append((tl / samples_per_epoch, vl))
        if VERBOSE:
            print("Epoch %d/%d - loss: %.6f - val_loss: %.6f" % (epoch + 1, nb_epoch, tl / samples_per_epoch, vl))
        if vl < best:  # ModelCheckpoint(save_best_only=True), train.py:109
            best = vl
            W.save_model(WEIGHTS_DIR, cfg, model.keras_list(), hp, wp)
    return history
