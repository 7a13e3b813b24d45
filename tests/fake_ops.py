"""A CPU stand-in for HipOps used ONLY by the CPU tests of the host logic (grouping,
Lloyd schedule, distributed exchange). It answers each C-ABI step with the oracle, so the
product's orchestration can run under gloo without a GPU. Never used by the product."""
import numpy as np
import torch

from oracle import spec_oracle as so


class OracleOps:
    def __init__(self, bank, device="cpu"):
        self.bank = bank
        self.device = torch.device(device)
        self.tapq = bank.tapq.astype(np.int64)
        self.calls = []

    # buffers are plain python holders
    def feature_slab(self, b, h, w):
        return {"x": None}

    def label_slab(self, b, h, w):
        return {"lab": None}

    def partial_slab(self, b, h, w, k):
        return {"sums": None}

    def new_centroids(self, n_sets, k):
        return torch.zeros((n_sets, k, self.bank.n_features), dtype=torch.int16)

    def new_sums(self, n_sets, k):
        return torch.zeros((n_sets, k, self.bank.n_features + 1), dtype=torch.int64)

    def gabor_features(self, imgs, feats):
        self.calls.append(("gabor", imgs.shape[0]))
        arr = imgs.numpy()
        feats["x"] = np.stack([so.gabor_features(im, self.tapq, self.bank.shift, self.bank.n_orient).reshape(self.bank.n_features, -1).T
                               for im in arr]).astype(np.int64)      # (B, P, D)

    def kmeans_init(self, feats, b, h, w, k, n_sets, cent):
        for s in range(n_sets):
            cent[s] = torch.from_numpy(so.kmeans_init(feats["x"][s], k).astype(np.uint16).view(np.int16))

    def features_gather(self, feats, b, h, w, byx):
        out = np.zeros((byx.shape[0], self.bank.n_features), np.uint16)
        for i, (bi, y, x) in enumerate(byx.numpy()):
            if bi >= 0:
                out[i] = feats["x"][bi][y * w + x]
        return torch.from_numpy(out.view(np.int16))

    def assign_accumulate(self, feats, cent, b, h, w, k, n_sets, labels, partials, rows=None, reverse=False):
        c = cent.numpy().view(np.uint16).astype(np.int64)
        lo, hi = rows if rows is not None else (0, h)
        vote = np.zeros((h, w), bool)
        vote[lo:hi] = True
        vote = vote.ravel()
        labs, sums = [], np.zeros((n_sets, k, self.bank.n_features + 1), np.int64)
        for i in range(b):
            s = i if n_sets == b else 0
            lab = so.kmeans_assign(feats["x"][i], c[s])
            _, cnt, sm = so.kmeans_update(feats["x"][i][vote], lab[vote], c[s])
            sums[s, :, :-1] += sm
            sums[s, :, -1] += cnt
            labs.append(lab)
        if labels is not None:
            labels["lab"] = np.stack(labs)
        if partials is not None:
            partials["sums"] = sums

    def reduce(self, partials, b, h, w, k, n_sets, sums):
        sums.copy_(torch.from_numpy(partials["sums"]))

    def finalize(self, sums, n_sets, k, cent):
        s = sums.numpy()
        c = cent.numpy().view(np.uint16).astype(np.int64)
        cnt = s[:, :, -1:]
        new = np.where(cnt > 0, (2 * s[:, :, :-1] + cnt) // np.maximum(2 * cnt, 1), c)
        cent.copy_(torch.from_numpy(new.astype(np.uint16).view(np.int16)))

    def reduce_finalize(self, partials, b, h, w, k, n_sets, sums, cent):
        self.reduce(partials, b, h, w, k, n_sets, sums)
        self.finalize(sums, n_sets, k, cent)

    def connected_regions(self, labels_i32, out):
        out.copy_(torch.from_numpy(np.stack([so.connected_regions(l) for l in labels_i32.numpy()])))

    def labels_widen(self, labels, b, h, w, out):
        out.copy_(torch.from_numpy(labels["lab"].reshape(b, h, w).astype(np.int32)))
