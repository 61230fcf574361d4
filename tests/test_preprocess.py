"""CLIP preprocessing: (CPU) the host-side coefficient tables drive a numpy emulation of the two-pass fixed-point resample
that equals PIL.Image.resize(BICUBIC) bit for bit; (GPU) the HIP kernels give exactly what PIL + the reference's transform
chain gives (Resize -> CenterCrop -> ToTensor -> Normalize), for down-, up- and no-scaling and odd sizes."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

SIZES = [(37, 53), (300, 500), (224, 224), (500, 375), (64, 64), (100, 31), (225, 224), (1, 900)]


def _img(h, w, seed):
    g = np.random.RandomState(seed)
    base = g.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    base[::2, ::3] = 255
    base[1::4] = 0
    return base


def _reference_transform(arr, n_px):
    """openai-CLIP `_transform` restated with PIL + numpy (oracle/preprocess.py)."""
    from oracle.preprocess import clip_transform
    return clip_transform(arr, n_px)


def _emulate(arr, oh, ow):
    import grip_amd  # noqa: F401
    from grip_amd.preprocess import resample_coeffs
    h, w = arr.shape[:2]
    def one_pass(a, in_size, out_size):          # a: [in_size, other, 3] -> [out_size, other, 3]
        coef, bounds, _ = resample_coeffs(in_size, out_size)
        out = np.zeros((out_size,) + a.shape[1:], dtype=np.uint8)
        for xx in range(out_size):
            lo, cnt = bounds[xx]
            acc = (a[lo:lo + cnt].astype(np.int64) * coef[xx, :cnt].astype(np.int64)[:, None, None]).sum(0) + (1 << 21)
            out[xx] = np.clip(acc >> 22, 0, 255)
        return out
    t = one_pass(arr.transpose(1, 0, 2), w, ow).transpose(1, 0, 2)       # horizontal first, 8-bit intermediate
    return one_pass(t, h, oh)


@pytest.mark.parametrize("h,w", SIZES[:6])
def test_coefficient_tables_reproduce_pillow_bicubic(h, w):
    import grip_amd  # noqa: F401
    from grip_amd.preprocess import resized_size
    arr = _img(h, w, h * 31 + w)
    oh, ow = resized_size(h, w, 32)
    want = np.asarray(Image.fromarray(arr).resize((ow, oh), Image.BICUBIC))
    assert np.array_equal(_emulate(arr, oh, ow), want)


@pytest.mark.parametrize("h,w", SIZES[:4])
def test_oracle_transform_accepts_arrays_and_pil_and_matches_the_emulation(h, w):
    import grip_amd  # noqa: F401
    from grip_amd.preprocess import MEAN, STD, resized_size
    arr = _img(h, w, h + 5 * w)
    a = _reference_transform(arr, 32)
    b = _reference_transform(Image.fromarray(arr), 32)
    assert a.shape == (3, 32, 32) and np.array_equal(a, b)
    oh, ow = resized_size(h, w, 32)
    px = _emulate(arr, oh, ow)
    top, left = int(round((oh - 32) / 2.0)), int(round((ow - 32) / 2.0))
    px = px[top:top + 32, left:left + 32].astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    want = (px - np.array(MEAN, np.float32)[:, None, None]) / np.array(STD, np.float32)[:, None, None]
    np.testing.assert_allclose(a, want, rtol=0, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("n_px", [224, 32])
def test_gpu_preprocess_matches_pil_transform(h, w, n_px):
    import grip_amd  # noqa: F401
    from grip_amd.preprocess import ClipPreprocess
    arr = _img(h, w, h + 7 * w)
    pre = ClipPreprocess(n_px, "cuda")
    got = pre(Image.fromarray(arr)).cpu().numpy()
    want = _reference_transform(arr, n_px)
    assert got.shape == (3, n_px, n_px)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)      # identical u8 pixels; float normalisation order only
    got2 = pre(torch.from_numpy(arr)).cpu().numpy()                # uint8 tensor input path
    assert np.array_equal(got, got2)


@pytest.mark.gpu
def test_clip_load_returns_the_gpu_preprocess_and_it_feeds_the_encoder():
    import grip_amd  # noqa: F401
    from grip_amd import clip
    m, preprocess = clip.load("small", device="cuda")
    ims = [Image.fromarray(_img(80 + 5 * i, 120 - 7 * i, i)) for i in range(3)]
    x = torch.stack([preprocess(im) for im in ims])
    assert x.shape == (3, 3, 64, 64) and x.is_cuda
    assert torch.isfinite(m.encode_image(x)).all()
    t = torch.randn(3, 64, 64)
    assert preprocess(t) is t                                      # already-preprocessed tensors pass through


@pytest.mark.gpu
@pytest.mark.parametrize("n_px", [224, 32])
def test_batched_preprocess_equals_per_image_and_pil(n_px, tmp_path):
    """grip_preprocess_batch (one packed upload + one launch pair for images of mixed sizes, incl. one that needs no resize) is
    bit-identical to the per-image kernels, hence to PIL's transform; load_batch decodes files on a thread pool first."""
    import grip_amd  # noqa: F401
    from grip_amd.preprocess import ClipPreprocess
    pre = ClipPreprocess(n_px, "cuda")
    sizes = SIZES + [(n_px, n_px), (n_px, 2 * n_px), (333, 77)]
    arrs = [_img(h, w, 3 * h + w) for h, w in sizes]
    single = torch.stack([pre(torch.from_numpy(a)) for a in arrs])
    batch = pre.batch([Image.fromarray(a) if i % 2 else a for i, a in enumerate(arrs)])
    assert batch.shape == (len(arrs), 3, n_px, n_px) and torch.equal(batch, single)
    for a, b in zip(arrs[:4], batch[:4]):
        np.testing.assert_allclose(b.cpu().numpy(), _reference_transform(a, n_px), rtol=0, atol=2e-6)
    paths = []
    for i, a in enumerate(arrs):
        p = tmp_path / f"im{i}.png"                 # lossless: the decoded pixels are exactly `a`
        Image.fromarray(a).save(p)
        paths.append(str(p))
    for workers in (1, 4):
        assert torch.equal(pre.load_batch(paths, workers=workers), single)
    assert pre.batch([]).shape == (0, 3, n_px, n_px)
    # decode PROCESSES around a shared-memory segment; a segment too small for the chunk (overflow images are uploaded one by one)
    # and a staging buffer that has to grow give the same tensor
    assert torch.equal(pre.load_batch(paths, processes=3), single)
    assert torch.equal(pre.load_batch(paths[::-1], processes=3), single.flip(0))
    pre.close()
    tight = ClipPreprocess(n_px, "cuda")
    tight.__dict__["_bytes_per_image"] = 16         # segment far too small at first: overflow uploads, then it grows
    try:
        for _ in range(4):
            assert torch.equal(tight.load_batch(paths, processes=2), single)
    finally:
        tight.close()
    small = ClipPreprocess(n_px, "cuda")
    small.__dict__["_bytes_per_image"] = 16
    assert torch.equal(small.load_batch(paths, workers=4), single)
    assert torch.equal(small.load_batch(paths, workers=4), single)
    # decode_chunk / finish_chunk split: chunk i+1 decoded on another thread while chunk i is uploaded
    import threading
    res = {}
    t = threading.Thread(target=lambda: res.setdefault("h", pre.decode_chunk(paths[3:], workers=2)))
    h0 = pre.decode_chunk(paths[:3], workers=2)
    t.start()
    first = pre.finish_chunk(h0)
    t.join()
    assert torch.equal(torch.cat([first, pre.finish_chunk(res["h"])]), single)


@pytest.mark.gpu
def test_pseudolabel_pool_from_image_files_uses_the_batched_loader(tmp_path, monkeypatch):
    """utils.pseudolabel_top_k on a dataset of image FILES (no pre-decoded pool): files are decoded on the thread pool and
    preprocessed by the batched kernel chunk by chunk; same lists as with the images preprocessed one by one up front."""
    import types

    import grip_amd  # noqa: F401
    from grip_amd import clip
    from grip_amd.utils import pseudolabel_top_k
    monkeypatch.chdir(tmp_path)
    m, preprocess = clip.load("small", device="cuda")
    g = np.random.RandomState(5)
    paths = []
    for i in range(23):
        a = np.clip(g.normal(128 + 40 * g.randn(3), 50, size=(70 + i, 90 - i, 3)), 0, 255).astype(np.uint8)
        p = tmp_path / f"{i:03d}.png"
        Image.fromarray(a).save(p)
        paths.append(str(p))
    classes = ["forest", "river", "highway"]
    l2i = {c: i for i, c in enumerate(classes)}
    cfg = types.SimpleNamespace(LEARNING_PARADIGM="ul", MODEL="textual_fpl")

    class DS:
        def __init__(self, images=None):
            self.filepaths, self.labels = list(paths), None
            if images is not None:
                self.images = images
    files = DS()
    pseudolabel_top_k(cfg, "Files", 3, "a photo of a {}", files, classes, preprocess, m, l2i, "cuda", "small", 1)
    pool = DS(torch.stack([preprocess(Image.open(p)) for p in paths]))
    pseudolabel_top_k(cfg, "Pool", 3, "a photo of a {}", pool, classes, preprocess, m, l2i, "cuda", "small", 1)
    assert (files.filepaths, files.labels) == (pool.filepaths, pool.labels) and len(files.filepaths) > 3


def test_decode_backends_pack_the_same_pixels(tmp_path):
    """data/decode.py (host only): thread and process back ends put every image's RGB pixels into the staging buffer exactly as
    Image.open(...).convert("RGB") gives them (JPEG, PNG, grey-scale and palette files); what does not fit comes back as overflow."""
    import grip_amd  # noqa: F401
    from grip_amd.data import decode as D
    shm_before = set(os.listdir("/dev/shm"))
    g = np.random.RandomState(3)
    paths = []
    for i in range(13):
        a = g.randint(0, 256, size=(20 + 3 * i, 50 - 2 * i, 3)).astype(np.uint8)
        im = Image.fromarray(a)
        if i % 4 == 1:
            im = im.convert("L")
        if i % 4 == 2:
            im = im.convert("P")
        p = tmp_path / (f"{i}.jpg" if i % 2 and i % 4 != 2 else f"{i}.png")
        im.save(p)
        paths.append(str(p))
    want = [np.asarray(Image.open(p).convert("RGB")) for p in paths]

    def check(buf, packed, overflow):
        for i, w in enumerate(want[:len(packed.shapes)]):
            assert tuple(packed.shapes[i]) == w.shape[:2]
            got = overflow[i] if i in overflow else buf[packed.offsets[i]:packed.offsets[i] + w.size].reshape(w.shape)
            assert np.array_equal(got, w), i
            assert packed.offsets[i] % D.ALIGN == 0
    buf = np.zeros(1 << 20, dtype=np.uint8)
    for pool in (None, D.make_thread_pool(4)):
        packed, overflow = D.decode_threads(paths, buf, pool)
        assert not overflow and packed.used <= buf.shape[0]
        check(buf, packed, overflow)
    tiny = np.zeros(9000, dtype=np.uint8)
    packed, overflow = D.decode_threads(paths, tiny, D.make_thread_pool(3))
    assert overflow and len(overflow) < len(paths)
    check(tiny, packed, overflow)
    dec = D.ProcessDecoder(3, 1 << 20, slots=2)
    try:
        for slot in (0, 1, 0):
            packed, overflow = dec.decode(paths, slot)
            assert not overflow
            check(dec.slot_view(slot), packed, overflow)
        assert len(dec.decode([], 0)[0].shapes) == 0
        with pytest.raises(RuntimeError, match="decode worker"):
            dec.decode(paths[:2] + [str(tmp_path / "missing.png")], 0)
        packed, overflow = dec.decode(paths[:5], 1)             # the workers survive a failed job
        check(dec.slot_view(1), packed, overflow)
        assert not dec.ensure(0, 1000) and dec.ensure(0, 3 << 20)      # slot 0 replaced by a larger segment: workers re-attach
        packed, overflow = dec.decode(paths, 0)
        assert not overflow
        check(dec.slot_view(0), packed, overflow)
        packed, overflow = dec.decode(paths[:1], 1)             # fewer images than workers
        check(dec.slot_view(1), packed, overflow)
    finally:
        dec.close()
    dec = D.ProcessDecoder(2, 20000, slots=1)
    try:
        packed, overflow = dec.decode(paths, 0)
        assert overflow
        check(dec.slot_view(0), packed, overflow)
    finally:
        dec.close()
    assert set(os.listdir("/dev/shm")) <= shm_before          # close() unlinked every segment


def test_worker_regions_are_compacted_into_the_staging_buffer():
    """The process back end's segment is 1.5x oversized and split evenly between the workers; the copy into the page-locked staging
    buffer (and therefore the upload) moves only what the workers wrote (data/decode.py Packed.compact_into)."""
    import grip_amd  # noqa: F401
    from grip_amd.data.decode import Packed
    r = np.random.RandomState(0)
    src = r.randint(0, 256, size=4096).astype(np.uint8)
    shapes = np.array([[2, 10], [3, 10], [1, 5], [4, 4]], dtype=np.int32)        # image bytes: 60, 90, 15, 48
    offsets = np.array([0, 256, 2048, 2304], dtype=np.int64)                    # worker 0 owns [0, 2048), worker 1 [2048, 4096)
    packed = Packed(offsets, shapes, 2304 + 256, regions=[(0, 512, 0, 2), (2048, 512, 2, 4)])
    dst = np.zeros(1024, dtype=np.uint8)
    out = packed.compact_into(src, dst)
    assert out.used == 1024 and out.offsets.tolist() == [0, 256, 512, 768]
    for i, (h, w) in enumerate(shapes):
        n = int(h) * int(w) * 3
        assert np.array_equal(dst[out.offsets[i]: out.offsets[i] + n], src[offsets[i]: offsets[i] + n])
