"""Image-decode kernel (uint8 HWC in HBM -> Pillow-exact bilinear resize -> float32 CHW) and the device loader on a
real MI355X, through the C ABI.  Integer work: bit-exact against PIL's own outputs and the CPU restatement."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import resize_oracle as RO
from mmdyn_hip import ops
from mmdyn_hip.utils import datasets as D
import synthetic_tree as ST

pytestmark = pytest.mark.gpu
DEV = "cuda"


def decode(frames_np, size, index=None):
    fr = torch.from_numpy(frames_np).to(DEV)
    dec = D.FrameDecoder(frames_np.shape[1], frames_np.shape[2], size, DEV)
    idx = torch.tensor(index, dtype=torch.int32, device=DEV) if index is not None else None
    return dec(fr, idx).cpu().numpy()


def as_chw(u8):
    return u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255)


def test_kernel_equals_pil_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "dataset_tree.npz"))
    np.testing.assert_array_equal(decode(g["resize/noise"][None], 64)[0], as_chw(g["resize/noise_64"]))
    np.testing.assert_array_equal(decode(g["resize/noise"][None], 128)[0], as_chw(g["resize/noise_128"]))
    np.testing.assert_array_equal(decode(g["resize/rect"][None], 64)[0], as_chw(g["resize/rect_64"]))      # 64 x 106
    np.testing.assert_array_equal(decode(g["tree/sample_visual"][None], (64, 64))[0], as_chw(g["tree/sample_visual_64"]))


@pytest.mark.parametrize("H,W,size", [(256, 256, 64), (256, 256, 256), (100, 80, 64), (64, 48, 96), (37, 53, (20, 70)),
                                      (512, 512, 32), (256, 256, (64, 256))])
def test_kernel_equals_oracle_shapes(H, W, size):
    """Down- and up-scaling, identity on one or both axes, odd sizes (unaligned rows take the byte-load path)."""
    rng = np.random.default_rng(H * 1000 + W)
    frames = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    frames[1] = 255                  # saturation: taps sum to 2^22 +- rounding, clip8 must hold 255
    frames[2, ::2] = 0
    out = decode(frames, size)
    oh, ow = D.resize_output_size(H, W, size)
    for i in range(3):
        np.testing.assert_array_equal(out[i], as_chw(RO.resize_bilinear_u8(frames[i], oh, ow)))


def test_gather_batch_and_full_size_properties():
    """BASELINE-sized decode: 1024 frames of 256x256x3 gathered out of a 96-frame store in one launch.  Properties:
    equal indices give equal outputs, a constant frame stays constant, outputs lie in [0,1] on the /255 lattice;
    plus exact equality with the oracle for a few of them."""
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (96, 256, 256, 3), dtype=np.uint8)
    frames[7] = 200
    index = rng.integers(0, 96, 1024)
    index[:4] = (7, 3, 3, 95)
    out = decode(frames, 64, index)
    assert out.shape == (1024, 3, 64, 64)
    np.testing.assert_array_equal(out[1], out[2])
    assert np.all(out[0] == np.float32(200) / np.float32(255))
    assert out.min() >= 0 and out.max() <= 1
    lattice = np.round(out * 255).astype(np.float32) / np.float32(255)
    np.testing.assert_array_equal(out, lattice)
    for b in (1, 3, 500, 1023):
        np.testing.assert_array_equal(out[b], RO.resize_to_tensor(frames[index[b]], 64))


def test_device_loader_and_training_on_a_tree(tmp_path):
    """dataset_setup on the synthetic PNG/json tree -> device loader -> one epoch of cnn-mvae seq_modeling."""
    import test_model_emu as T
    from mmdyn_hip.problems.problems import SeqModeling
    ST.build_tree(str(tmp_path))
    random.seed(7)
    out = D.dataset_setup(str(tmp_path), "seq_modeling", input_size=(64, 64), batchsize=2, shuffle=False)
    ds = out["train_dataset"]
    data, target = next(iter(out["train_loader"]))
    assert data[0].is_cuda and tuple(data[0].shape) == (6, 3, 64, 64)
    np.testing.assert_array_equal(data[1][4].cpu().numpy(), RO.resize_to_tensor(ds.data[1][1][1], 64))
    np.testing.assert_array_equal(target[3][5].cpu().numpy(), RO.resize_to_tensor(ds.targets[1][2][3], 64))
    prob = SeqModeling(T.args(num_epochs=1, batchsize=2, dataset_path=str(tmp_path)), log_dir=str(tmp_path / "log"))
    # the pickle exists by now, so like the reference the problem sees seq_length None ([::None]: every frame)
    assert prob._seq_length is None and len(prob.train_loader) == 2
    prob.train()
    assert os.path.exists(os.path.join(str(tmp_path / "log"), "results.pkl"))
    # dyn_modeling (one-step predictor) on the same tree: flat frame axis, every frame a sample
    from mmdyn_hip.problems.problems import DynModeling
    dyn = DynModeling(T.args(problem_type="dyn_modeling", num_epochs=1, batchsize=2, dataset_path=str(tmp_path)),
                      log_dir=str(tmp_path / "dyn"))
    d, t = next(iter(dyn.train_loader))
    assert tuple(d[0].shape) == (6, 3, 64, 64) and dyn._seq_length == 3      # read off the data (the pickle exists)
    dyn.train()
    assert dyn._step.last["means"].shape[0] == 6
