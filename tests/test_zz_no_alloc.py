"""ptamd_filter_list_static (the entry point that allocates nothing) against
ptamd_filter_list, through the C ABI, bit for bit: two calls in a row, the second on a
smaller list; a list over the static limit is refused."""
import numpy as np
import pytest

import pathtracer_amd as pt
from pathtracer_amd import cabi
from conftest import scene_path

pytestmark = pytest.mark.gpu


def _survivors(dev, fn, pixels, n, n_done, state, torch):
    out = torch.full((max(n, 1),), -7, dtype=torch.int32, device=pixels.device)
    n_out = torch.full((1,), -1, dtype=torch.int64, device=pixels.device)
    n_dev = torch.tensor([n], dtype=torch.int64, device=pixels.device)
    lst = cabi.PixelList(pixels.data_ptr(), n_dev.data_ptr(), 0, n, 0, 1)
    fn(lst, n_done, state.data_ptr(), out.data_ptr(), n_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return int(n_out.item()), out.cpu().numpy()


def test_static_filter_equals_allocating_filter():
    import torch
    w, h = 320, 240
    dev = pt.Device(0)
    sc = pt.Scene(scene_path("CBspheres_lambertian"))
    dev.upload(sc)
    dev.set_camera(sc.camera(w, h))
    p = cabi.make_params(w, h, ns_aa=128, max_ray_depth=3, ns_area_light=2, seed=13)
    device = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    npix = w * h
    pixels = torch.arange(npix, dtype=torch.int32, device=device)
    dev.frame_begin(p)
    # a state in which every fifth pixel has converged after 32 samples (zero variance) and the
    # others have not (mean 1, variance ~1): rows are s1, s2, sum.rgb, n
    st = torch.zeros((npix, 6), dtype=torch.float64, device=device)
    st[:, 0] = 32.0
    st[:, 1] = 64.0
    st[::5, 1] = 32.0
    st[:, 2:5] = 32.0
    st[:, 5] = 32.0
    state = st.reshape(-1).contiguous()
    # first call: the whole frame; second call: a smaller list (every third pixel)
    small = pixels[1::3].contiguous()
    for lst, n in ((pixels, npix), (small, small.numel())):
        ka, a = _survivors(dev, dev.filter_list, lst, n, 32, state, torch)
        kb, b = _survivors(dev, dev.filter_list_static, lst, n, 32, state, torch)
        expect = lst.cpu().numpy()
        expect = expect[expect % 5 != 0]
        assert 0 < ka < n and ka == expect.size
        assert ka == kb
        assert np.array_equal(a[:ka], expect) and np.array_equal(a[:ka], b[:kb])
        assert (b[kb:] == -7).all()            # nothing written past the survivors
    # and once more on the big list: what the smaller call left in the scratch array does not matter
    ka, a = _survivors(dev, dev.filter_list, pixels, npix, 32, state, torch)
    kb, b = _survivors(dev, dev.filter_list_static, pixels, npix, 32, state, torch)
    assert ka == kb and np.array_equal(a[:ka], b[:kb])
    # over the limit: refused with a clear error, nothing launched
    too_many = 4 * 1024 * 1024 + 1
    big = cabi.PixelList(pixels.data_ptr(), 0, too_many, too_many, 0, 1)
    with pytest.raises(pt.PtamdError, match="PTAMD_FILTER_STATIC_MAX_PIXELS"):
        dev.filter_list_static(big, 32, state.data_ptr(), pixels.data_ptr(), state.data_ptr(), stream)
    dev.close()
