"""Deterministic miniature of the reference's on-disk dataset tree (exp_1_flat_plane.py:128-155 /
exp_3_force_pert.py:139-142): <root>/dataset/<synset>/<obj>/sequence_NNNN/{visual,tactile,seg}_NNNN.png + data.json.
Used by tests/golden/make_golden.py (to run the reference's compiler on it) and by the tests (to run ours)."""
import json
import os

import numpy as np
from PIL import Image


def digest(a):
    import hashlib
    a = np.ascontiguousarray(a)
    return hashlib.sha256(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes()).hexdigest()


def build_tree(root, n_seq=8, seq_len=3, h=96, w=80, shock=True, seed=11):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    for s in range(n_seq):
        d = os.path.join(root, "dataset", "synset_a" if s % 2 == 0 else "synset_b", f"obj{s // 2}", f"sequence_{s:04d}")
        os.makedirs(d, exist_ok=True)
        info = {"time_step": list(range(seq_len)), "time": [0.1 * t for t in range(seq_len)],
                "position": rng.uniform(-0.5, 0.5, (seq_len, 3)).tolist(),
                "orientation": rng.uniform(-1, 1, (seq_len, 4)).tolist()}
        if shock:
            info["shock"] = rng.uniform(-3, 3, (seq_len, 3)).tolist()
        with open(os.path.join(d, "data.json"), "w") as f:
            json.dump(info, f)
        for t in range(seq_len):
            cy, cx = rng.integers(20, h - 20), rng.integers(20, w - 20)
            ry, rx = rng.integers(6, 18), rng.integers(6, 18)
            seg = np.zeros((h, w), dtype=np.uint8)
            seg[5:15, 5:25] = 1                                        # the id the reference maps to background
            seg[(np.abs(yy - cy) <= ry) & (np.abs(xx - cx) <= rx)] = 2 + (s % 3)
            visual = (rng.integers(0, 256, (h, w, 3)) // 4 + np.stack([yy, xx, yy + xx], -1) // 2).astype(np.uint8)
            tactile = rng.integers(0, 256, (h, w, 3), dtype=np.uint8) if (s + t) % 4 else np.full((h, w, 3), 77, np.uint8)
            Image.fromarray(visual).save(os.path.join(d, f"visual_{t:04d}.png"))
            Image.fromarray(tactile).save(os.path.join(d, f"tactile_{t:04d}.png"))
            Image.fromarray(seg).save(os.path.join(d, f"seg_{t:04d}.png"))
    return root


def describe(compiled):
    """Order-preserving digest of a compiled {'data','targets'} dict: one sha256 per array."""
    out = {}
    for name in ("data", "targets"):
        for s, seq in enumerate(compiled[name]):
            for t, frame in enumerate(seq):
                for j, a in enumerate(frame):
                    out[f"{name}/{s}/{t}/{j}"] = digest(np.asarray(a))
    return out
