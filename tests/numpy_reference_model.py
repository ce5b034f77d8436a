"""A third, independent statement of the aligner half of the path: numpy, written from the reference's source lines (cited), sharing no code
with oracle/ -- used by tests/test_gpu_parity.py to check the GPU without the oracle in between.  Integer-valued results (index images,
correspondence lists) are produced with the same fp32 operations as the reference and must match the GPU exactly; the least-squares sums are
float64 (the bar for them is a tolerance).  Test infrastructure only."""
import numpy as np

f32 = np.float32


def roundf(a):
    """C roundf: half away from zero, exactly"""
    t = np.trunc(a); fr = a - t
    return np.where(fr >= f32(0.5), t + 1, np.where(fr <= f32(-0.5), t - 1, t)).astype(np.float32)


def project(P, KRt, min_d, max_d, rows, cols):
    """PinholePointProjector::project (pinholepointprojector.cpp:33-66, .h:224-233): index + depth image; nearest point per pixel, ties keep the
    lower index (the sequential loop's strict '>').  P [n,3] float32, KRt row-major 4x4 float32."""
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    def row(r): return ((KRt[r, 0] * x + KRt[r, 1] * y) + KRt[r, 2] * z) + KRt[r, 3] * f32(1.0)
    ix, iy, d = row(0), row(1), row(2)
    ok = ~((d < f32(min_d)) | (d > f32(max_d)))
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = f32(1.0) / d
        rx, ry = roundf(ix * inv), roundf(iy * inv)
        ok &= (rx >= 0) & (rx < cols) & (ry >= 0) & (ry < rows)
    idx = np.nonzero(ok)[0]
    pix = ry[idx].astype(np.int64) * cols + rx[idx].astype(np.int64)
    order = np.lexsort((idx, d[idx], pix))
    pix_s, idx_s = pix[order], idx[order]
    first = np.ones(len(pix_s), bool); first[1:] = pix_s[1:] != pix_s[:-1]
    wi = np.full(rows * cols, -1, np.int32); wd = np.full(rows * cols, np.finfo(np.float32).max, np.float32)
    wi[pix_s[first]] = idx_s[first]; wd[pix_s[first]] = d[idx_s[first]]
    return wi.reshape(rows, cols), wd.reshape(rows, cols)


def _sq3(a):
    return (a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1]) + a[:, 2] * a[:, 2]


def _iso(T, p, w):
    """Isometry3f * Vector4f with left-to-right fp32 sums"""
    return np.stack([((T[k, 0] * p[:, 0] + T[k, 1] * p[:, 1]) + T[k, 2] * p[:, 2]) + T[k, 3] * f32(w) for k in range(3)], 1)


def correspondences(ref, cur, ref_index, cur_index, T, normal_thr, dist_thr, flat_thr, ratio_thr):
    """CorrespondenceFinder::compute (correspondencefinder.cpp:45-106), one thread: (list [C,2] in row-major pixel order, K)"""
    r_i, c_i = ref_index.reshape(-1), cur_index.reshape(-1)
    cand = (r_i >= 0) & (c_i >= 0)
    rI, cI = r_i[cand], c_i[cand]
    rP, rN, cP, cN = ref["points"][rI], ref["normals"][rI], cur["points"][cI], cur["normals"][cI]
    ok = (_sq3(cN) != 0) & (_sq3(rN) != 0)
    rp, rn = _iso(T, rP, 1.0), _iso(T, rN, 0.0)
    ok &= ~(((cN[:, 0] * rn[:, 0] + cN[:, 1] * rn[:, 1]) + cN[:, 2] * rn[:, 2]) < f32(normal_thr))
    ok &= ~(_sq3(cP[:, :3] - rp) > f32(dist_thr) * f32(dist_thr))
    rc = np.maximum(ref["curvature"][rI], f32(flat_thr)); cc = np.maximum(cur["curvature"][cI], f32(flat_thr))
    ratio = ((rc.astype(np.float64) + 1e-5) / (cc.astype(np.float64) + 1e-5)).astype(np.float32)
    mx = f32(ratio_thr); mn = f32(1.0) / mx
    ok &= ~((ratio < mn) | (ratio > mx))
    return np.stack([rI[ok], cI[ok]], 1).astype(np.int32), int(cand.sum())


def _skew(v):
    """bm_se3.h:54-66: S = -2 [v]x  (n, 3, 3)"""
    S = np.zeros((len(v), 3, 3))
    S[:, 0, 1] = 2 * v[:, 2]; S[:, 1, 0] = -2 * v[:, 2]
    S[:, 0, 2] = -2 * v[:, 1]; S[:, 2, 0] = 2 * v[:, 1]
    S[:, 1, 2] = 2 * v[:, 0]; S[:, 2, 1] = -2 * v[:, 0]
    return S


def linearize(ref, cur, corr, invT, max_chi2, robust=True):
    """Linearizer::update (linearizer.cpp:33-114) in float64 from the fp32 clouds: H, b, chi2, inliers"""
    T = np.asarray(invT, np.float64)
    ri, ci = corr[:, 0], corr[:, 1]
    rp = ref["points"][ri, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    rn = ref["normals"][ri, :3].astype(np.float64) @ T[:3, :3].T
    pe = rp - cur["points"][ci, :3]; ne = rn - cur["normals"][ci, :3]
    oP = cur["omega_p"][ci].reshape(-1, 4, 4).transpose(0, 2, 1)[:, :3, :3].astype(np.float64)
    oN = cur["omega_n"][ci].reshape(-1, 4, 4).transpose(0, 2, 1)[:, :3, :3].astype(np.float64)
    ep = np.einsum("nij,nj->ni", oP, pe); en = np.einsum("nij,nj->ni", oN, ne)
    le = (pe * ep).sum(1) + (ne * en).sum(1)
    k = np.ones(len(le))
    over = le > max_chi2
    if robust:
        k[over] = np.sqrt(max_chi2 / le[over]); keep = np.ones(len(le), bool)
    else:
        keep = ~over
    Sp, Sn = _skew(rp), _skew(rn)
    Htt = oP[keep].sum(0)
    Htr = np.einsum("nij,njk->ik", oP[keep], Sp[keep])
    Hrr = np.einsum("nji,njk,nkl->il", Sp[keep], oP[keep], Sp[keep]) + np.einsum("nji,njk,nkl->il", Sn[keep], oN[keep], Sn[keep])
    bt = (k[keep, None] * ep[keep]).sum(0)
    br = (k[keep, None] * (np.einsum("nji,nj->ni", Sp[keep], ep[keep]) + np.einsum("nji,nj->ni", Sn[keep], en[keep]))).sum(0)
    H = np.zeros((6, 6)); H[:3, :3] = Htt; H[:3, 3:] = Htr; H[3:, 3:] = Hrr; H[3:, :3] = Htr.T
    return H, np.concatenate([bt, br]), float((k[keep] * le[keep]).sum()), int(keep.sum())


# ---------------------------------------------------------------------------------------------------------------- projector matrices, converter
def _mm3(A, B):
    """3x3 (or 3x3 . 3xk) fp32 product with left-to-right inner products"""
    A = np.asarray(A, np.float32); B = np.asarray(B, np.float32)
    out = np.empty((3, B.shape[1]), np.float32)
    for i in range(3):
        for j in range(B.shape[1]):
            out[i, j] = f32(f32(f32(A[i, 0] * B[0, j]) + f32(A[i, 1] * B[1, j])) + f32(A[i, 2] * B[2, j]))
    return out


def inverse3(m):
    """Eigen's Matrix3f::inverse() (compute_inverse<.., 3>: cofactors of the first column, det = (c0 * m00 + c1 * m10) + c2 * m20, the adjugate
    times 1 / det), every operation fp32"""
    m = np.asarray(m, np.float32)
    def cof(i, j):
        i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
        return f32(f32(m[i1, j1] * m[i2, j2]) - f32(m[i1, j2] * m[i2, j1]))
    c0 = [cof(0, 0), cof(1, 0), cof(2, 0)]
    det = f32(f32(f32(c0[0] * m[0, 0]) + f32(c0[1] * m[1, 0])) + f32(c0[2] * m[2, 0]))
    invdet = f32(1.0) / det
    r = np.empty((3, 3), np.float32)
    for i in range(3):
        for j in range(3):
            r[i, j] = f32(cof(j, i) * invdet)
    return r


def projector_matrices(K, T):
    """PinholePointProjector::_updateMatrices (pinholepointprojector.cpp:17-31): KRt = [K R' | K t'] with (R', t') = inverse(T) = (R^T, -R^T t);
    iKRt = [R iK | t].  K = (fx, fy, cx, cy)."""
    fx, fy, cx, cy = [f32(k) for k in K]
    Km = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32)
    T = np.asarray(T, np.float32)
    Rt = T[:3, :3].T.copy()
    tp = _mm3(-Rt, T[:3, 3:4])
    KRt = np.eye(4, dtype=np.float32); KRt[:3, :3] = _mm3(Km, Rt); KRt[:3, 3:4] = _mm3(Km, tp)
    iK = inverse3(Km)                                                  # _iK = _cameraMatrix.inverse(): Eigen's 3x3 cofactor inverse
    iKRt = np.eye(4, dtype=np.float32); iKRt[:3, :3] = _mm3(T[:3, :3], iK); iKRt[:3, 3] = T[:3, 3]
    return KRt, iKRt, iK


def unproject(depth, iKRt, min_d, max_d):
    """PinholePointProjector::unProject (pinholepointprojector.cpp:93-133, .h:246-251): (valid mask, x, y, z images); a valid pixel's point index is
    its row-major rank"""
    rows, cols = depth.shape
    valid = ~((depth < f32(min_d)) | (depth > f32(max_d)))
    cc, rr = np.meshgrid(np.arange(cols, dtype=np.float32), np.arange(rows, dtype=np.float32))
    a, b, d = cc * depth, rr * depth, depth
    def row(k): return ((iKRt[k, 0] * a + iKRt[k, 1] * b) + iKRt[k, 2] * d) + iKRt[k, 3] * f32(1.0)
    x, y, z = [np.where(valid, row(k), f32(0)).astype(np.float32) for k in range(3)]
    return valid, x, y, z


def integral_planes(valid, x, y, z):
    """PointIntegralImage::compute (pointintegralimage.cpp:7-44): the ten distinct sums, sequential fp32 prefix along image x, then along image y"""
    planes = [x, y, z, valid.astype(np.float32), x * x, x * y, x * z, y * y, y * z, z * z]
    return [np.cumsum(np.cumsum(p.astype(np.float32), axis=1, dtype=np.float32), axis=0, dtype=np.float32) for p in planes]


def intervals(depth, valid, K, world_radius):
    """projectIntervals (pinholepointprojector.cpp:135-147, .h:264-274): int(max(fx R / d, fy R / d)) with p = K (R, R, 0) * (1 / d)"""
    fx, fy = f32(K[0]), f32(K[1]); R = f32(world_radius)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = f32(1.0) / np.where(valid, depth, f32(1.0))
        px, py = (fx * R) * inv, (fy * R) * inv
        return np.where(valid, np.where(px > py, px, py).astype(np.int32), -1)


def window_sums(I, itv, valid, min_radius, max_radius):
    """getRegion (pointintegralimage.cpp:53-66) for every valid pixel in row-major order: the ten sums ((A + B) - C) - D in that order"""
    rows, cols = valid.shape
    rr, cc = np.nonzero(valid)
    rad = np.clip(itv[rr, cc], min_radius, max_radius)
    xmin, xmax = np.clip(cc - rad - 1, 0, cols - 1), np.clip(cc + rad - 1, 0, cols - 1)
    ymin, ymax = np.clip(rr - rad - 1, 0, rows - 1), np.clip(rr + rad - 1, 0, rows - 1)
    return [((Ik[ymax, xmax] + Ik[ymin, xmin]) - Ik[ymax, xmin]) - Ik[ymin, xmax] for Ik in I]


def mean_and_covariance(v):
    """PointAccumulator::mean / covariance (pointaccumulator.h:66-86), fp32: (n, mean[3], the six lower-triangle covariance entries)"""
    with np.errstate(divide="ignore", invalid="ignore"):
        d = f32(1.0) / v[3]
        mean = [v[k] * d for k in range(3)]
        cov = {(0, 0): v[4] * d - mean[0] * mean[0], (1, 0): v[5] * d - mean[1] * mean[0], (2, 0): v[6] * d - mean[2] * mean[0],
               (1, 1): v[7] * d - mean[1] * mean[1], (2, 1): v[8] * d - mean[2] * mean[1], (2, 2): v[9] * d - mean[2] * mean[2]}
    return v[3].astype(np.int32), mean, cov


# ---------------------------------------------------------------------------------------------------------------- depth helpers, matchClouds score
def depth_16u_to_32f(raw, scale=0.001):
    """DepthImage_convert_16UC1_to_32FC1 (pwn_static.cpp:54-68): scale * raw where raw != 0, else 0"""
    raw = np.asarray(raw, np.uint16)
    return np.where(raw != 0, f32(scale) * raw.astype(np.float32), f32(0)).astype(np.float32)


def depth_32f_to_16u(img, scale=1000.0):
    """DepthImage_convert_32FC1_to_16UC1 (pwn_static.cpp:38-52): (unsigned short)(scale * f) where f < FLT_MAX, else 0"""
    img = np.asarray(img, np.float32)
    fin = img < np.finfo(np.float32).max
    return np.where(fin, (f32(scale) * np.where(fin, img, f32(0))).astype(np.int64), 0).astype(np.uint16)


def depth_scale(src, step, max_depth_cov=0.01):
    """DepthImage_scale (pwn_static.cpp:5-36): block mean of ALL pixels of a step x step block over the count of its positive ones, dropped when
    acc2 / np - mu * mu > maxDepthCov; fp32 accumulation in the loop's order (i over rows, j over columns)"""
    src = np.asarray(src, np.float32)
    rows, cols = src.shape[0] // step, src.shape[1] // step
    acc = np.zeros((rows, cols), np.float32); acc2 = np.zeros((rows, cols), np.float32); cnt = np.zeros((rows, cols), np.int32)
    for i in range(step):
        for j in range(step):
            blk = src[i:rows * step:step, j:cols * step:step][:rows, :cols]
            acc = acc + blk; acc2 = acc2 + blk * blk; cnt += (blk > 0)
    out = np.zeros((rows, cols), np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        npf = cnt.astype(np.float32)
        mu = acc / npf
        sigma = acc2 / npf - mu * mu
    ok = (cnt > 0) & ~(sigma > f32(max_depth_cov))
    out[ok] = mu[ok]
    return out


def match_score(ref_depth, cur_depth, threshold=50.0):
    """PwnMatcherBase::matchClouds' depth-agreement score (pwn_tracker/pwn_matcher_base.cpp:153-182): both finder depth images to uint16 millimetres,
    mask = both > 0, diff = abs(cur - ref) BITWISE-ANDed with the float mask 255.0f (what cv::Mat's operator& does on CV_32F data), inliers = masked
    and diff < threshold, sum over ALL pixels in row-major order (fp32), distance = sum / nonZeros"""
    cur = depth_32f_to_16u(cur_depth).astype(np.float32); ref = depth_32f_to_16u(ref_depth).astype(np.float32)
    mask = (cur > 0) & (ref > 0)
    ad = np.abs(cur - ref).astype(np.float32)
    bits = ad.view(np.uint32) & np.where(mask, np.uint32(0x437F0000), np.uint32(0))
    diff = bits.view(np.float32)
    non_zeros = int(mask.sum())
    inliers = int((mask & (diff < f32(threshold))).sum())
    total = np.cumsum(diff.reshape(-1), dtype=np.float32)[-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        dist = float(f32(total) / f32(non_zeros))
    return dict(image_nonZeros=non_zeros, image_inliers=inliers, image_outliers=non_zeros - inliers, image_reprojectionDistance=dist)


# ---------------------------------------------------------------------------------------------------------------- 3x3 symmetric eigen-solve
def eig3_direct(c00, c10, c20, c11, c21, c22):
    """Eigen::SelfAdjointEigenSolver<Matrix3f>::computeDirect(A, ComputeEigenvectors) (SelfAdjointEigenSolver.h, direct_selfadjoint_eigenvalues<.,3,false>:
    computeRoots, extract_kernel, run), vectorised over n matrices given by their lower triangles, every arithmetic operation fp32 in the source's order.
    The three libm calls (atan2, cos, sin on floats) are taken as correctly rounded: float64 numpy, rounded once.  Returns (evals [n,3] ascending,
    U [n,3,3] with the eigenvectors in columns)."""
    a = [np.asarray(v, np.float32) for v in (c00, c10, c20, c11, c21, c22)]
    c00, c10, c20, c11, c21, c22 = a
    n = len(c00)
    eps = np.finfo(np.float32).eps
    with np.errstate(all="ignore"):
        shift = ((c00 + c11) + c22) / f32(3.0)
        m00, m11, m22 = c00 - shift, c11 - shift, c22 - shift
        m10, m20, m21 = c10.copy(), c20.copy(), c21.copy()
        scale = np.maximum.reduce([np.abs(m00), np.abs(m11), np.abs(m22), np.abs(m10), np.abs(m20), np.abs(m21)])
        pos = scale > 0
        sdiv = np.where(pos, scale, f32(1.0))
        m00, m11, m22, m10, m20, m21 = [np.where(pos, v / sdiv, v).astype(np.float32) for v in (m00, m11, m22, m10, m20, m21)]
        # computeRoots
        s_inv3 = f32(1.0) / f32(3.0); s_sqrt3 = np.sqrt(f32(3.0))
        c0 = m00 * m11 * m22 + f32(2.0) * m10 * m20 * m21 - m00 * m21 * m21 - m11 * m20 * m20 - m22 * m10 * m10
        c1 = m00 * m11 - m10 * m10 + m00 * m22 - m20 * m20 + m11 * m22 - m21 * m21
        c2 = m00 + m11 + m22
        c2_3 = c2 * s_inv3
        a_3 = np.maximum((c2 * c2_3 - c1) * s_inv3, f32(0.0))
        half_b = f32(0.5) * (c0 + c2_3 * (f32(2.0) * c2_3 * c2_3 - c1))
        q = np.maximum(a_3 * a_3 * a_3 - half_b * half_b, f32(0.0))
        rho = np.sqrt(a_3)
        theta = (np.arctan2(np.sqrt(q).astype(np.float64), half_b.astype(np.float64)).astype(np.float32)) * s_inv3
        cos_t = np.cos(theta.astype(np.float64)).astype(np.float32); sin_t = np.sin(theta.astype(np.float64)).astype(np.float32)
        e0 = c2_3 - rho * (cos_t + s_sqrt3 * sin_t)
        e1 = c2_3 - rho * (cos_t - s_sqrt3 * sin_t)
        e2 = c2_3 + f32(2.0) * rho * cos_t
        ev = np.stack([e0, e1, e2], 1).astype(np.float32)

        def kernel(d00, d11, d22):
            """extract_kernel of the matrix with diagonal (d00, d11, d22) and off-diagonals (m10, m20, m21): (res, representative)"""
            cols = [np.stack([d00, m10, m20], 1), np.stack([m10, d11, m21], 1), np.stack([m20, m21, d22], 1)]
            i0 = np.zeros(n, np.int64); best = np.abs(d00)
            up = np.abs(d11) > best; i0[up] = 1; best = np.where(up, np.abs(d11), best)
            up = np.abs(d22) > best; i0[up] = 2
            C = np.stack(cols, 1)                                     # [n, col, component]
            idx = np.arange(n)
            rep, n1, n2 = C[idx, i0], C[idx, (i0 + 1) % 3], C[idx, (i0 + 2) % 3]
            def cross(u, v): return np.stack([u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1], u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2], u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]], 1)
            x0, x1 = cross(rep, n1), cross(rep, n2)
            s0, s1 = _sq3(x0), _sq3(x1)
            first = s0 > s1
            res = np.where(first[:, None], x0 / np.sqrt(s0)[:, None], x1 / np.sqrt(s1)[:, None]).astype(np.float32)
            return res, rep

        d0 = e2 - e1; d1 = e1 - e0
        swapped = d0 > d1
        d0 = np.where(swapped, d1, d0)
        ek = np.where(swapped, e2, e0); el = np.where(swapped, e0, e2)
        vk, vl = kernel(m00 - ek, m11 - ek, m22 - ek)
        tiny = d0 <= f32(2.0) * eps * d1
        dd = (vk[:, 0] * vl[:, 0] + vk[:, 1] * vl[:, 1]) + vk[:, 2] * vl[:, 2]
        vo = vl - dd[:, None] * vl
        vo = vo / np.sqrt(_sq3(vo))[:, None]
        vl2, _ = kernel(m00 - el, m11 - el, m22 - el)
        vl = np.where(tiny[:, None], vo, vl2).astype(np.float32)
        v0 = np.where(swapped[:, None], vl, vk); v2 = np.where(swapped[:, None], vk, vl)
        v1 = np.stack([v2[:, 1] * v0[:, 2] - v2[:, 2] * v0[:, 1], v2[:, 2] * v0[:, 0] - v2[:, 0] * v0[:, 2], v2[:, 0] * v0[:, 1] - v2[:, 1] * v0[:, 0]], 1)
        z = _sq3(v1)
        v1 = np.where((z > 0)[:, None], v1 / np.sqrt(np.where(z > 0, z, f32(1.0)))[:, None], v1).astype(np.float32)
        U = np.stack([v0, v1, v2], 2).astype(np.float32)            # [n, component, column]
        ident = (e2 - e0) <= eps
        U[ident] = np.eye(3, dtype=np.float32)
        ev = ev * scale[:, None]
        ev = ev + shift[:, None]
    return ev.astype(np.float32), U


# ---------------------------------------------------------------------------------------------------------------- the whole converter
def convert(depth, K, conv, point_flat=(1000.0, 1.0, 1.0), normal_flat=100.0, normal_nonflat=1.0):
    """DepthImageConverterIntegralImage::compute (depthimageconverterintegralimage.cpp:15-55) with an identity sensor offset: unProject +
    projectIntervals, PointIntegralImage::compute, StatsCalculatorIntegralImage::compute (statscalculatorintegralimage.cpp:33-80), the two information
    matrix calculators (informationmatrixcalculator.cpp:9-58).  conv = the keyword set of g2o_frontend_amd/conf.py.  Returns a dict of arrays in the
    layout of the oracle's / the GPU cloud's arrays() (points / normals [M,4], curvature [M], omega_p / omega_n [M,16] column-major 4x4, eigenvalues
    [M,3], npoints [M]) plus the index and interval images."""
    depth = np.asarray(depth, np.float32)
    rows, cols = depth.shape
    _, iKRt, _ = projector_matrices(K, np.eye(4, dtype=np.float32))
    valid, x, y, z = unproject(depth, iKRt, conv["min_distance"], conv["max_distance"])
    Mn = int(valid.sum())
    index = np.full((rows, cols), -1, np.int32); index[valid] = np.arange(Mn, dtype=np.int32)
    itv = intervals(depth, valid, K, conv["world_radius"])
    I = integral_planes(valid, x, y, z)
    v = window_sums(I, itv, valid, conv["min_image_radius"], conv["max_image_radius"])
    n, mean, cov = mean_and_covariance(v)
    has = n >= conv["min_points"]
    P = np.stack([x[valid], y[valid], z[valid]], 1)
    ev = np.zeros((Mn, 3), np.float32); U = np.tile(np.eye(3, dtype=np.float32), (Mn, 1, 1))
    sel = np.nonzero(has)[0]
    if len(sel):
        e, u = eig3_direct(*[cov[k][sel] for k in ((0, 0), (1, 0), (2, 0), (1, 1), (2, 1), (2, 2))])
        e[:, 0] = np.maximum(e[:, 0], f32(0.0))                                                   # statscalculatorintegralimage.cpp:66-67
        ev[sel] = e; U[sel] = u
    with np.errstate(all="ignore"):
        curv = (ev[:, 0].astype(np.float64) / ((ev[:, 0] + ev[:, 1] + ev[:, 2]).astype(np.float32).astype(np.float64) + 1e-9)).astype(np.float32)     # stats.h:98-103
    normals = np.zeros((Mn, 4), np.float32)
    keep = has & (curv < f32(conv["stats_curvature_threshold"]))                                  # :72-78
    n0 = U[:, :, 0].copy()
    dotp = ((n0[:, 0] * P[:, 0] + n0[:, 1] * P[:, 1]) + n0[:, 2] * P[:, 2]) + f32(0.0) * f32(1.0)   # 4-vector dot, w = 0 * 1
    n0[dotp > 0] = -n0[dotp > 0]
    normals[keep, :3] = n0[keep]
    curvature = np.where(has, curv, f32(0.0)).astype(np.float32)                                  # Stats() default: eigenvalues 0 -> curvature 0 / 1e-9 = 0
    nonzero = _sq3(normals) > 0
    flat = curvature < f32(conv["point_info_curvature_threshold"])
    with np.errstate(all="ignore"):
        dg = np.where(flat[:, None], np.asarray(point_flat, np.float32)[None, :], f32(1.0) / ev).astype(np.float32)      # informationmatrixcalculator.cpp:22-30
        om = np.zeros((Mn, 3, 3), np.float32)
        for i in range(3):
            for j in range(3):
                om[:, i, j] = ((U[:, i, 0] * dg[:, 0]) * U[:, j, 0] + (U[:, i, 1] * dg[:, 1]) * U[:, j, 1]) + (U[:, i, 2] * dg[:, 2]) * U[:, j, 2]
    om[~nonzero] = 0
    omega_p = np.zeros((Mn, 4, 4), np.float32); omega_p[:, :3, :3] = om
    on = np.where((curvature < f32(conv["normal_info_curvature_threshold"]))[:, None, None], np.eye(3, dtype=np.float32) * f32(normal_flat),
                  np.eye(3, dtype=np.float32) * f32(normal_nonflat)).astype(np.float32)             # :47-56
    on[~nonzero] = 0
    omega_n = np.zeros((Mn, 4, 4), np.float32); omega_n[:, :3, :3] = on
    pts = np.ones((Mn, 4), np.float32); pts[:, :3] = P
    return dict(points=pts, normals=normals, curvature=curvature, omega_p=omega_p.transpose(0, 2, 1).reshape(Mn, 16).copy(),
                omega_n=omega_n.transpose(0, 2, 1).reshape(Mn, 16).copy(), eigenvalues=ev, npoints=np.where(has, n, 0).astype(np.int32),
                index=index, interval=itv)


# ---------------------------------------------------------------------------------------------------------------- Aligner::_computeStatistics
def _quat2mat(q):
    qx, qy, qz = q; qw = np.sqrt(max(0.0, 1.0 - q @ q))
    return np.array([[qw * qw + qx * qx - qy * qy - qz * qz, 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
                     [2 * (qx * qy + qz * qw), qw * qw - qx * qx + qy * qy - qz * qz, 2 * (qy * qz - qx * qw)],
                     [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), qw * qw - qx * qx - qy * qy + qz * qz]])


def _v2t(x):
    X = np.eye(4); X[:3, 3] = x[:3]; X[:3, :3] = _quat2mat(np.asarray(x[3:6], float)); return X


def _t2v(X):
    """bm_se3.h:45-52: translation + vector part of the normalised quaternion with w >= 0"""
    R = X[:3, :3]; t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0); w = 0.5 * s; s = 0.5 / s
        q = np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s])
    else:
        i = int(np.argmax(np.diag(R))); j = (i + 1) % 3; k = (j + 1) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = np.zeros(3); q[i] = 0.5 * s; s = 0.5 / s
        w = (R[k, j] - R[j, k]) * s; q[j] = (R[j, i] + R[i, j]) * s; q[k] = (R[k, i] + R[i, k]) * s
    nrm = np.sqrt(q @ q + w * w); q = q / nrm; w = w / nrm
    return np.concatenate([X[:3, 3], -q if w < 0 else q])


def compute_statistics(H, T):
    """Aligner::_computeStatistics (aligner.cpp:152-199) from the linearizer's H and the final transform, float64: H + I, its inverse as the local
    covariance, the unscented sigma points of unscented.h:23-47 (alpha 1e-3, beta 2) remapped by p -> t2v(T v2t(p)^-1), the Gaussian reconstructed
    from them (unscented.h:49-65), Omega = covariance^-1 and the singular-value ratios of its two diagonal 3x3 blocks."""
    H = np.asarray(H, np.float64) + np.eye(6); T = np.asarray(T, np.float64)
    sigma = np.linalg.inv(H)
    dim = 6; alpha, beta = 1e-3, 2.0
    lam = alpha * alpha * dim
    wi = 1.0 / (2.0 * (dim + lam))
    L = np.linalg.cholesky(sigma * (dim + lam))
    pts = [(np.zeros(6), lam / (dim + lam), lam / (dim + lam) + (1.0 - alpha * alpha + beta))]
    for i in range(dim):
        pts.append((L[:, i].copy(), wi, wi)); pts.append((-L[:, i], wi, wi))
    samples = [(_t2v(T @ np.linalg.inv(_v2t(p))), a, b) for p, a, b in pts]
    mean = sum(a * s for s, a, _ in samples)
    cov = sum(b * np.outer(s - mean, s - mean) for s, _, b in samples)
    omega = np.linalg.inv(cov)
    sv_t = np.linalg.svd(omega[:3, :3], compute_uv=False); sv_r = np.linalg.svd(omega[3:, 3:], compute_uv=False)
    return dict(mean=mean, omega=omega, translationalEigenRatio=sv_t[0] / sv_t[2], rotationalEigenRatio=sv_r[0] / sv_r[2])


# ---------------------------------------------------------------------------------------------------------------- SE(3) priors
def prior_terms(kind, mean, information, invT, reference_transform=None):
    """SE3RelativePrior / SE3AbsolutePrior (se3_prior.cpp:8-71) as Aligner::align uses them (aligner.cpp:96-108), float64: error(invT) =
    t2v(invT * [R^-1 *] mean); jacobian = central differences of error(v2t(+-eps e_i) * invT), eps = 1e-3; errorInformation = Jz^-T Omega Jz^-1 with
    Jz the central differences of the error with the mean moved to mean * v2t(+-eps e_i).  Returns (Hp, bp) = (J^T I' J, J^T I' e)."""
    mean = np.asarray(mean, np.float64); info = np.asarray(information, np.float64); invT = np.asarray(invT, np.float64)
    pre = np.eye(4) if kind == 0 else np.linalg.inv(np.asarray(reference_transform, np.float64))
    def err(iT, mu): return _t2v(iT @ pre @ mu)
    eps = 1e-3
    J = np.zeros((6, 6)); Jz = np.zeros((6, 6))
    for i in range(6):
        up = np.zeros(6); up[i] = eps
        J[:, i] = (0.5 / eps) * (err(_v2t(up) @ invT, mean) - err(_v2t(-up) @ invT, mean))
        Jz[:, i] = (0.5 / eps) * (err(invT, mean @ _v2t(up)) - err(invT, mean @ _v2t(-up)))
    iJz = np.linalg.inv(Jz)
    Ir = iJz.T @ info @ iJz
    e = err(invT, mean)
    return J.T @ Ir @ J, J.T @ Ir @ e
