"""CPU: hand-checkable micro-cases that anchor the UNPINNED third-party restatements (OpenCV ellipse dilation, spconv
active-set rules) in oracle/region.py, and cross-check the two independent sparse-conv restatements
(gather tables in oracle/refmodel.py vs dense-masked oracle/standins/spconv_standin.py)."""
import numpy as np
import torch

from oracle import region, refmodel
from oracle.standins import spconv_standin as sp


def test_ellipse_known_answers():
    assert region.ellipse_kernel(5).tolist() == [[0, 0, 1, 0, 0], [1] * 5, [1] * 5, [1] * 5, [0, 0, 1, 0, 0]]   # well-known OpenCV 5x5
    assert region.ellipse_kernel(1).tolist() == [[1]]
    assert region.ellipse_kernel(3).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    k15 = region.ellipse_kernel(15)
    assert k15.shape == (15, 15) and k15[7].sum() == 15 and k15[0].sum() == 1 and np.array_equal(k15, k15[::-1]) and np.array_equal(k15, k15.T[::-1].T)


def test_dilate_matches_bruteforce_all_widths():
    rs = np.random.RandomState(0)
    img = (rs.uniform(size=(23, 31)) > 0.93).astype(np.uint8)
    img[0, 0] = 1
    img[-1, -1] = 1
    for k in range(1, 30):
        assert np.array_equal(region.dilate(img, k), region.dilate_bruteforce(img, k)), k


def test_dilate_matches_scipy_ndimage_all_widths():
    """A third, independent implementation: scipy.ndimage.maximum_filter with the ellipse as footprint is the same correlation-form
    dilation with the anchor at k // 2 and "pixels outside the image do not count" (mode='constant', cval=0) that cv2.dilate documents."""
    import scipy.ndimage as ndi
    rs = np.random.RandomState(3)
    img = (rs.uniform(size=(37, 41)) > 0.95).astype(np.uint8)
    img[0, 0] = img[-1, -1] = img[0, -1] = 1
    for k in range(1, 30):
        ref = ndi.maximum_filter(img, footprint=region.ellipse_kernel(k).astype(bool), mode='constant', cval=0)
        assert np.array_equal(region.dilate(img, k), ref), k


def test_compute_unknown_thresholds_and_single_pixel():
    a = np.zeros((1, 32, 32), np.float32)
    a[0, 16, 16] = 0.5
    a[0, 2, 2] = 1.0 / 255.0          # NOT unknown (strict inequality)
    a[0, 3, 3] = 254.0 / 255.0        # NOT unknown
    out = region.compute_unknown(a, 30, False)
    exp = np.zeros((32, 32), np.uint8)
    exp[16 - 7:16 + 8, 16 - 7:16 + 8] = region.ellipse_kernel(15)
    assert np.array_equal(out[0], exp)


def test_active_pyramid_rules():
    roi = np.zeros((1, 16, 16), np.uint8)
    roi[0, 5, 6] = 1                   # i = 2*o - 1 + k: y=5 -> o in {2 (k=2), 3 (k=0)}; x=6 -> o = 3 (k=1) only (even)
    a1, a2, a4, a8 = region.active_pyramid(roi)
    assert sorted(map(tuple, np.argwhere(a2[0]))) == [(2, 3), (3, 3)]
    assert a4.shape == (1, 4, 4) and a8.shape == (1, 2, 2)
    inv = region.inverse_neighbors(a1, a2)
    rows = region.index_grid(a2)[0]
    assert inv.shape == (1, 9)
    assert inv[0, 2 * 3 + 1] == rows[2, 3] and inv[0, 0 * 3 + 1] == rows[3, 3] and (inv[0] >= 0).sum() == 2


def test_gather_restatement_equals_dense_masked_standin():
    rs = np.random.RandomState(1)
    act = rs.uniform(size=(2, 16, 16)) > 0.7
    co = torch.from_numpy(region.coords_of(act))
    feat = torch.from_numpy(rs.normal(size=(co.shape[0], 8)).astype(np.float32))
    x = sp.SparseConvTensor(feat, co, (16, 16), 2)
    subm = sp.SubMConv2d(8, 6, 3, padding=1, bias=True, indice_key='s')
    down = sp.SparseConv2d(8, 8, 3, stride=2, padding=1, bias=False, indice_key='d')
    inv = sp.SparseInverseConv2d(8, 5, 3, bias=False, indice_key='d')
    with torch.no_grad():
        subm.bias.normal_()
        y = subm(x)
        ref = refmodel._gather_conv(feat, region.subm_neighbors(act), subm.weight, subm.bias)
        assert torch.allclose(y.features, ref, atol=1e-5)
        c = down(x)
        a2 = region.downsample_active(act)
        assert np.array_equal(c.indices.numpy(), region.coords_of(a2))
        up = inv(c)
        ref_up = refmodel._gather_conv(c.features, region.inverse_neighbors(act, a2), inv.weight)
        assert np.array_equal(up.indices.numpy(), region.coords_of(act))
        assert torch.allclose(up.features, ref_up, atol=1e-5)
