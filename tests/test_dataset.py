"""The dataset reader in front of the hot path, CPU side: the Pillow-resampler
restatement against vectors produced by PIL itself, the host-side plan of the C library, the tree compiler against the
pickle the REFERENCE's compiler wrote for the same synthetic tree, and the device loader through the emulation."""
import os
import pickle
import random

import numpy as np
import pytest
import torch

from oracle import resize_oracle as RO
from mmdyn_hip import ops
from mmdyn_hip.utils import datasets as D
import synthetic_tree as ST
from emu_backend import EmuBackend


@pytest.fixture()
def emu():
    prev = ops.B
    ops.set_backend(EmuBackend())
    yield
    ops.set_backend(prev)


def g_load(golden_dir):
    return np.load(os.path.join(golden_dir, "dataset_tree.npz"), allow_pickle=False)


def test_resize_oracle_matches_pil_vectors(golden_dir):
    g = g_load(golden_dir)
    np.testing.assert_array_equal(RO.resize_bilinear_u8(g["resize/noise"], 64, 64), g["resize/noise_64"])
    np.testing.assert_array_equal(RO.resize_bilinear_u8(g["resize/noise"], 128, 128), g["resize/noise_128"])
    assert RO.resize_output_size(120, 200, 64) == (64, 106)
    np.testing.assert_array_equal(RO.resize_bilinear_u8(g["resize/rect"], 64, 106), g["resize/rect_64"])
    np.testing.assert_array_equal(RO.resize_bilinear_u8(g["tree/sample_visual"], 64, 64), g["tree/sample_visual_64"])
    t = RO.resize_to_tensor(g["resize/noise"], 64)
    assert t.dtype == np.float32 and t.shape == (3, 64, 64)
    np.testing.assert_array_equal(t, g["resize/noise_64"].transpose(2, 0, 1).astype(np.float32) / np.float32(255))


@pytest.mark.parametrize("a,b", [(256, 64), (256, 128), (120, 64), (200, 106), (50, 100), (256, 37), (1000, 7)])
def test_library_plan_equals_oracle(a, b):
    bo, co = ops.HipBackend().resize_plan(a, b, "cpu")
    rb, rc = RO.precompute_coeffs(a, b)
    np.testing.assert_array_equal(bo.numpy(), rb)
    np.testing.assert_array_equal(co.numpy(), rc)
    assert int(co.sum(1).min()) > 0 and abs(int(co.sum(1).max()) - (1 << 22)) <= co.shape[1]


@pytest.mark.parametrize("tag,shock", [("shock", True), ("plain", False)])
def test_tree_compiler_writes_the_reference_pickle(golden_dir, tmp_path, tag, shock):
    g = g_load(golden_dir)
    ST.build_tree(str(tmp_path), shock=shock)
    random.seed(7)
    ds = D.VisuoTactileDataset(train=True, dataset_path=str(tmp_path))
    with open(ds.dataset_path, "rb") as f:
        comp = pickle.load(f)
    desc = ST.describe(comp)
    assert list(desc.keys()) == [str(k) for k in g[f"tree/{tag}/keys"]]
    assert list(desc.values()) == [str(k) for k in g[f"tree/{tag}/sha"]]
    assert len(ds) == int(g[f"tree/{tag}/n_train"]) and ds.seq_length == int(g[f"tree/{tag}/seq_length"])
    test = D.VisuoTactileDataset(train=False, dataset_path=str(tmp_path))
    assert len(test) == int(g[f"tree/{tag}/n_test"]) and test.seq_length is None      # loaded, not compiled: sic
    assert ds.shock_dim == (3 if shock else 0)
    if shock:
        np.testing.assert_array_equal(np.asarray(comp["data"][0][1][2]), g["tree/sample_pose"])
        np.testing.assert_array_equal(np.asarray(comp["data"][0][1][4]), g["tree/sample_shock"])


def test_device_loader_batches(golden_dir, tmp_path, emu):
    ST.build_tree(str(tmp_path))
    random.seed(7)
    out = D.dataset_setup(str(tmp_path), "seq_modeling", input_size=(64, 64), batchsize=2, shuffle=False, device="cpu")
    ds, loader = out["train_dataset"], out["train_loader"]
    assert set(out) == {"train_dataset", "test_dataset", "train_loader", "test_loader", "seq_length"}
    L = out["seq_length"]
    assert L == 3 and len(loader) == len(ds) // 2 and loader.shock_dim == 3
    batches = list(loader)
    assert len(batches) == len(loader)
    data, target = batches[0]
    assert [tuple(x.shape) for x in data] == [(2 * L, 3, 64, 64), (2 * L, 3, 64, 64), (2 * L, 7), (2 * L, 2), (2 * L, 3)]
    assert [tuple(x.shape) for x in target] == [(2 * L, 3, 64, 64), (2 * L, 3, 64, 64), (2 * L, 7), (2 * L, 3, 64, 64)]
    for s in range(2):
        for t in range(L):
            row = s * L + t
            np.testing.assert_array_equal(data[0][row].numpy(), RO.resize_to_tensor(ds.data[s][t][0], 64))
            np.testing.assert_array_equal(data[1][row].numpy(), RO.resize_to_tensor(ds.data[s][t][1], 64))
            np.testing.assert_array_equal(data[2][row].numpy(), ds.data[s][t][2].astype(np.float32))
            np.testing.assert_array_equal(data[4][row].numpy(), ds.data[s][t][4].astype(np.float32))
            np.testing.assert_array_equal(target[0][row].numpy(), RO.resize_to_tensor(ds.targets[s][t][0], 64))
            np.testing.assert_array_equal(target[2][row].numpy(), ds.targets[s][t][2].astype(np.float32))
            np.testing.assert_array_equal(target[3][row].numpy(), RO.resize_to_tensor(ds.targets[s][t][3], 64))
    # the final frame of a sequence is one object in the pickle and one slot in the store
    st = ds.store("cpu")
    assert st["frames"].shape[0] < 2 * 4 * L * len(ds) and st["frames"].dtype == torch.uint8
    # default collation keeps the frame axis; a per-sample read returns [L, ...]
    d2, _ = ds.batch([0, 1], "cpu", fold=False)
    assert tuple(d2[0].shape) == (2, L, 3, 64, 64)
    np.testing.assert_array_equal(d2[0].reshape(2 * L, 3, 64, 64).numpy(), data[0].numpy())
    item_d, item_t = ds[1]
    np.testing.assert_array_equal(item_d[0].numpy(), data[0][L:2 * L].numpy())
    # shuffling permutes whole sequences
    sh = D.DeviceLoader(ds, 2, shuffle=True, device="cpu", seed=3)
    assert len(list(sh)) == len(sh)


def test_device_loader_seeding_and_rank_sharding(tmp_path):
    """Shuffling follows a seed (None: drawn from torch's global generator, so runs differ like the reference's
    DataLoader; a default-constructed torch.Generator would repeat one order for ever), and with world_size > 1 the ranks
    take disjoint shares of the same permuted global mini-batches."""
    class Toy:
        shock_dim = 0

        def __len__(self):
            return 12

        def batch(self, idx, device, fold):
            return torch.as_tensor(idx).clone()

    a = [b.tolist() for b in D.DeviceLoader(Toy(), 2, shuffle=True, device="cpu", seed=5)]
    b = [b.tolist() for b in D.DeviceLoader(Toy(), 2, shuffle=True, device="cpu", seed=5)]
    c = [b.tolist() for b in D.DeviceLoader(Toy(), 2, shuffle=True, device="cpu", seed=6)]
    assert a == b and a != c and sorted(sum(a, [])) == list(range(12))
    torch.manual_seed(123)
    d = [b.tolist() for b in D.DeviceLoader(Toy(), 2, shuffle=True, device="cpu")]
    e = [b.tolist() for b in D.DeviceLoader(Toy(), 2, shuffle=True, device="cpu")]
    assert d != e                                            # unseeded loaders differ from one another
    shards = [[b.tolist() for b in D.DeviceLoader(Toy(), 2, shuffle=True, device="cpu", seed=9, rank=r, world_size=3)]
              for r in range(3)]
    assert all(len(s) == 2 for s in shards)                  # 12 // (2 * 3) global mini-batches
    seen = sum((sum(s, []) for s in shards), [])
    assert len(set(seen)) == len(seen) == 12
    whole = [b.tolist() for b in D.DeviceLoader(Toy(), 6, shuffle=True, device="cpu", seed=9)]
    for step in range(2):                                    # the ranks' shares tile the single-process batch, in rank order
        assert sum((shards[r][step] for r in range(3)), []) == whole[step]
