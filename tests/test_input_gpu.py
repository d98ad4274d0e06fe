"""GPU input pipeline (csrc/image.hip, vision_amd/input_pipeline.py) against the transform the reference's data
module applies on the host (reference dataloader.py:32-36): Pillow's bilinear Image.resize + ToTensor.  Bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

SIZES = [(37, 53, 16, 16), (300, 200, 224, 224), (224, 224, 224, 224), (100, 224, 224, 224), (224, 100, 224, 224),
         (512, 640, 224, 224), (16, 16, 32, 48), (448, 7, 64, 64), (225, 223, 224, 224), (2, 3, 5, 4), (1024, 1024, 1024, 512),
         (1365, 2048, 1024, 1024)]


def _tables(n_in, n_out):
    from vision_amd._lib import call, lib
    ks = lib.vamd_resize_ksize(n_in, n_out)
    b = np.empty((n_out, 2), dtype=np.int32)
    c = np.empty((n_out, ks), dtype=np.int32)
    call("vamd_resize_coeffs", n_in, n_out, b.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), ks)
    return torch.from_numpy(b).to(DEV), torch.from_numpy(c).to(DEV), ks


@pytest.mark.parametrize("pixel_bytes", [3, 4])
@pytest.mark.parametrize("shape", SIZES, ids=lambda s: "%dx%d-%dx%d" % s)
def test_resize_kernel_equals_pillow(shape, pixel_bytes):
    from vision_amd._lib import call
    H, W, Ho, Wo = shape
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    img[: H // 2, : W // 3] = 255
    img[H // 2:, : W // 4] = 0
    want = np.asarray(Image.fromarray(img, "RGB").resize((Wo, Ho), Image.BILINEAR))
    if pixel_bytes == 4:                                   # Pillow's own storage: a pad byte after every pixel
        img = np.concatenate([img, rng.integers(0, 256, (H, W, 1), dtype=np.uint8)], axis=2)
    src = torch.from_numpy(img).to(DEV)
    ht = _tables(W, Wo) if W != Wo else None
    vt = _tables(H, Ho) if H != Ho else None
    tmp = torch.empty(H * Wo * 3, dtype=torch.uint8, device=DEV)
    out_u8 = torch.full((Ho, Wo, 3), 7, dtype=torch.uint8, device=DEV)
    out_f = torch.full((3, Ho, Wo), 7.0, device=DEV)
    P = lambda t: t.data_ptr() if t is not None else None
    call("vamd_image_resize_u8", P(src), H, W, pixel_bytes, P(ht[0]) if ht else None, P(ht[1]) if ht else None, ht[2] if ht else 0,
         P(vt[0]) if vt else None, P(vt[1]) if vt else None, vt[2] if vt else 0, P(tmp), Ho, Wo, P(out_f), P(out_u8),
         torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(out_u8.cpu().numpy(), want)
    to_tensor = torch.from_numpy(want.copy()).permute(2, 0, 1).float().div(255)      # transforms.ToTensor
    assert torch.equal(out_f.cpu(), to_tensor)


def _write_images(root, n, seed=0):
    rng = np.random.default_rng(seed)
    for c in ("a", "b"):
        os.makedirs(os.path.join(root, c), exist_ok=True)
    paths = []
    for i in range(n):
        h, w = int(rng.integers(20, 90)), int(rng.integers(20, 90))
        if i == 0:
            h, w = 32, 32                                    # already at the patch size: convert only
        if i == 1:
            h, w = 5, 400                                    # elongated beyond the pipeline's limit: Pillow resizes it
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        p = os.path.join(root, "ab"[i % 2], f"img{i:03d}.png")
        (Image.fromarray(img, "RGB") if i % 3 else Image.fromarray(img[:, :, 0], "L")).save(p)   # some greyscale files
        paths.append(p)
    return paths


def test_pipeline_batches_equal_host_transform(tmp_path):
    from vision_amd.input_pipeline import GpuImagePipeline
    paths = _write_images(str(tmp_path), 23)
    pipe = GpuImagePipeline(32, DEV, workers=4, prefetch=2)
    chunks = [paths[i:i + 5] for i in range(0, len(paths), 5)]
    host = lambda p: torch.from_numpy(np.asarray(Image.open(p).convert("RGB").resize((32, 32), Image.BILINEAR)).copy()) \
        .permute(2, 0, 1).float().div(255)
    got = list(pipe.batches(chunks))
    assert [g.shape[0] for g in got] == [5, 5, 5, 5, 3]
    for chunk, x in zip(chunks, got):
        assert torch.equal(x.cpu(), torch.stack([host(p) for p in chunk]))
    assert torch.equal(pipe.load_batch(chunks[1]).cpu(), got[1].cpu())
    # a consumer that stops early leaves no thread behind
    it = pipe.batches(chunks)
    next(it)
    it.close()
    pipe.close()


def test_datamodule_gpu_transform_equals_pil_path(tmp_path):
    """ImageFolderDataModule (reference dataloader.py:10-82) with the transform on the GPU yields the same batches,
    labels included, as with the transform on the host."""
    from vision_amd.trainer import ImageFolderDataModule
    _write_images(str(tmp_path), 31, seed=1)
    out = {}
    for mode in (True, False):
        torch.manual_seed(5)
        dm = ImageFolderDataModule(str(tmp_path), patch_size=32, batch_size=4, use_manual_split=True, device=DEV,
                                   gpu_transform=mode, num_workers=3)
        dm.setup()
        out[mode] = [list(dm.train_dataloader()), list(dm.val_dataloader()), list(dm.test_dataloader())]
    for a, b in zip(out[True], out[False]):
        assert len(a) == len(b) and len(a) > 0
        for (xa, ya), (xb, yb) in zip(a, b):
            assert xa.is_cuda and xa.dtype == torch.float32 and torch.equal(xa, xb) and torch.equal(ya, yb)
