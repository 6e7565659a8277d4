"""ORACLE (test infrastructure, not product): integer / boolean region ops of the MaGGIe hot path.

CPU numpy restatement of
  * `compute_unknown`            -- /root/reference/maggie/utils/utils.py:27-55
  * the OpenCV structuring element + binary dilation it calls (third-party `opencv-python`,
    UNPINNED in /root/reference/requirements.txt:3, not vendored => "parity unpinned" vs real cv2):
    restated from OpenCV's published algorithm (`getStructuringElement(MORPH_ELLIPSE)`,
    `dilate` with default anchor (k/2,k/2) and default border = "ignore outside pixels").
  * the active-site pyramid that spconv's `SparseConv2d(k=3,s=2,p=1)` rule-book generation produces
    (/root/reference/maggie/network/decoder/resnet_inst_matt_spconv.py:61-66,217-218); third-party
    `spconv-cu120`, UNPINNED (requirements.txt:4) => "parity unpinned" vs real spconv.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

LOWER_THRES = 1.0 / 255.0
UPPER_THRES = 254.0 / 255.0


def ellipse_kernel(k: int) -> np.ndarray:
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k)) as a (k,k) uint8 array.

    OpenCV algorithm: r = k//2 (rows), c = k//2 (cols); row i has ones on [c-dx, c+dx] with
    dx = cvRound(c * sqrt((r^2 - (i-r)^2) / r^2)) when |i-r| <= r, else the row is empty.
    k == 1 degenerates to a 1x1 rectangle. cvRound is round-half-to-even (np.rint).
    """
    if k == 1:
        return np.ones((1, 1), np.uint8)
    r = k // 2
    c = k // 2
    inv_r2 = 1.0 / (float(r) * r) if r else 0.0
    elem = np.zeros((k, k), np.uint8)
    for i in range(k):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
            j1 = max(c - dx, 0)
            j2 = min(c + dx + 1, k)
            elem[i, j1:j2] = 1
    return elem


def ellipse_row_spans(k: int):
    """Per-row [lo, hi] column offsets (relative to the anchor k//2) of the ellipse; None for empty rows.

    Returns a list of (dy, lo, hi) with dy, lo, hi relative to the anchor.
    """
    elem = ellipse_kernel(k)
    a = k // 2
    spans = []
    for i in range(k):
        nz = np.nonzero(elem[i])[0]
        if nz.size:
            # rows of the OpenCV ellipse are contiguous runs
            assert nz[-1] - nz[0] + 1 == nz.size
            spans.append((i - a, int(nz[0]) - a, int(nz[-1]) - a))
    return spans


def dilate(img: np.ndarray, k: int) -> np.ndarray:
    """cv2.dilate(img, ellipse(k)) for a 2-D uint8 image (values 0/1), default anchor & border.

    dst(y,x) = max over SE offsets (dy,dx) of src(y+dy, x+dx), pixels outside the image ignored.
    """
    h, w = img.shape
    out = np.zeros_like(img)
    src = img.astype(bool)
    for dy, lo, hi in ellipse_row_spans(k):
        # rows of dst that read src row y+dy
        y0 = max(0, -dy)
        y1 = min(h, h - dy)
        if y1 <= y0:
            continue
        srow = src[y0 + dy:y1 + dy]                       # (rows, w)
        # horizontal run-OR over offsets [lo, hi] via prefix sums
        cs = np.zeros((srow.shape[0], w + 1), np.int32)
        np.cumsum(srow, axis=1, out=cs[:, 1:])
        xs = np.arange(w)
        a = np.clip(xs + lo, 0, w)
        b = np.clip(xs + hi + 1, 0, w)
        hit = (cs[:, b] - cs[:, a]) > 0
        out[y0:y1] |= hit.astype(out.dtype)
    return out


def dilate_bruteforce(img: np.ndarray, k: int) -> np.ndarray:
    """Definition-level dilation used to cross-check `dilate` in tests (slow, tiny images only)."""
    elem = ellipse_kernel(k)
    a = k // 2
    h, w = img.shape
    out = np.zeros_like(img)
    for y in range(h):
        for x in range(w):
            v = 0
            for i in range(k):
                for j in range(k):
                    if elem[i, j]:
                        yy, xx = y + i - a, x + j - a
                        if 0 <= yy < h and 0 <= xx < w and img[yy, xx]:
                            v = 1
            out[y, x] = v
    return out


def unknown_widths(n_slices: int, k_size: int, is_train: bool) -> np.ndarray:
    """Structuring-element width per slice (utils.py:45-49). Train mode draws from the GLOBAL numpy RNG,
    one `np.random.randint(1, k_size)` per slice in slice order, exactly like the reference."""
    if is_train:
        return np.array([np.random.randint(1, k_size) for _ in range(n_slices)], np.int32)
    return np.full((n_slices,), k_size // 2, np.int32)


def compute_unknown(masks: np.ndarray, k_size: int = 30, is_train: bool = False, widths=None) -> np.ndarray:
    """utils.py:28-55 on a float array (..., h, w) -> uint8 array of the same shape.

    uncertain = (m > 1/255) & (m < 254/255) in float32, then per (h,w) slice a dilation with the
    ellipse of width `k_size//2` (eval) or a random width in [1, k_size) (train).
    """
    m = np.asarray(masks, np.float32)
    h, w = m.shape[-2:]
    unc = ((m > np.float32(LOWER_THRES)) & (m < np.float32(UPPER_THRES))).astype(np.uint8)
    flat = unc.reshape(-1, h, w)
    if widths is None:
        widths = unknown_widths(flat.shape[0], k_size, is_train)
    out = np.empty_like(flat)
    for n in range(flat.shape[0]):
        out[n] = dilate(flat[n], int(widths[n]))
    return out.reshape(unc.shape)


# ----------------------------------------------------------------------------------------------
# Active-site pyramid (spconv SparseConv2d(k=3, s=2, p=1) output-site rule), all dense boolean maps
# ----------------------------------------------------------------------------------------------

def downsample_active(act: np.ndarray) -> np.ndarray:
    """Output sites of SparseConv2d(kernel 3, stride 2, padding 1): (..., H, W) bool -> (..., Ho, Wo).

    Ho = (H + 2 - 3)//2 + 1. Output site o is active iff any active input i = 2*o - 1 + k, k in {0,1,2}.
    """
    act = np.asarray(act).astype(bool)
    H, W = act.shape[-2:]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    pad = np.zeros(act.shape[:-2] + (2 * Ho + 1, 2 * Wo + 1), bool)
    pad[..., 1:H + 1, 1:W + 1] = act
    out = np.zeros(act.shape[:-2] + (Ho, Wo), bool)
    for ky in range(3):
        for kx in range(3):
            out |= pad[..., ky:ky + 2 * Ho:2, kx:kx + 2 * Wo:2]
    return out


def active_pyramid(roi: np.ndarray):
    """roi (B, H, W) {0,1} -> [A1, A2, A4, A8] boolean maps (the 'detail' index pyramid)."""
    a1 = np.asarray(roi) > 0
    a2 = downsample_active(a1)
    a4 = downsample_active(a2)
    a8 = downsample_active(a4)
    return [a1, a2, a4, a8]


def coords_of(act: np.ndarray) -> np.ndarray:
    """Row-major sorted (batch, y, x) int32 coordinates of active sites == torch.nonzero order
    (resnet_inst_matt_spconv.py:206-214)."""
    return np.ascontiguousarray(np.argwhere(act).astype(np.int32))


def index_grid(act: np.ndarray) -> np.ndarray:
    """(B,H,W) int32 grid: row id of each active site in `coords_of` order, -1 elsewhere."""
    grid = np.full(act.shape, -1, np.int32)
    grid[act] = np.arange(int(act.sum()), dtype=np.int32)
    return grid


def subm_neighbors(act: np.ndarray, ksize: int = 3) -> np.ndarray:
    """Neighbour table of a submanifold conv: (N, k*k) int32, entry = row of site (y+ky-c, x+kx-c) or -1."""
    grid = index_grid(act)
    B, H, W = act.shape
    c = ksize // 2
    pad = np.full((B, H + 2 * c, W + 2 * c), -1, np.int32)
    pad[:, c:c + H, c:c + W] = grid
    co = coords_of(act)
    nbr = np.empty((co.shape[0], ksize * ksize), np.int32)
    for ky in range(ksize):
        for kx in range(ksize):
            nbr[:, ky * ksize + kx] = pad[co[:, 0], co[:, 1] + ky, co[:, 2] + kx]
    return nbr


def inverse_neighbors(act_fine: np.ndarray, act_coarse: np.ndarray) -> np.ndarray:
    """Gather table of SparseInverseConv2d(k=3) reusing the (fine -> coarse) pairs of a
    SparseConv2d(k=3,s=2,p=1): (N_fine, 9) int32. Entry k=(ky,kx) of fine site i is the coarse row o
    with i = 2*o - 1 + k (per axis), or -1 when the parity does not match / o is out of range.
    (Every such o is active by construction: i is active and lies in o's receptive field.)"""
    grid = index_grid(act_coarse)
    B, Hc, Wc = act_coarse.shape
    co = coords_of(act_fine)
    nbr = np.full((co.shape[0], 9), -1, np.int32)
    for ky in range(3):
        ty = co[:, 1] + 1 - ky
        oky = (ty % 2 == 0) & (ty >= 0) & (ty // 2 < Hc)
        for kx in range(3):
            tx = co[:, 2] + 1 - kx
            ok = oky & (tx % 2 == 0) & (tx >= 0) & (tx // 2 < Wc)
            oy = np.where(ok, ty // 2, 0)
            ox = np.where(ok, tx // 2, 0)
            rows = grid[co[:, 0], oy, ox]
            nbr[:, ky * 3 + kx] = np.where(ok, rows, -1)
    return nbr
