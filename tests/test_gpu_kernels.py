"""GPU parity of the primitive kernels (through the C ABI) against numpy/scipy
on the same seeded inputs.  Elementwise results and CSR row sums must be
bit-identical (the kernels are built with -ffp-contract=off and sum rows left
to right like scipy's csr_matvec); reductions agree to rounding."""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dv():
    from ipsolver import device
    return device


@pytest.mark.parametrize("n", [1, 7, 64, 1000, 65537, 1 << 20])
def test_elementwise_bitexact(dv, n):
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    X, Y = dv.DVec.from_host(x), dv.DVec.from_host(y)
    assert np.array_equal((X + Y).to_host(), x + y)
    assert np.array_equal((X - Y).to_host(), x - y)
    assert np.array_equal((X * Y).to_host(), x * y)
    assert np.array_equal((-X).to_host(), -x)
    assert np.array_equal((2.5 * X).to_host(), 2.5 * x)
    assert np.array_equal((X + 0.3 * Y).to_host(), x + 0.3 * y)
    assert np.array_equal((X + 1.25).to_host(), x + 1.25)
    assert np.array_equal((1.0 - X).to_host(), 1.0 - x)
    lb, ub = np.full(n, -0.5), np.full(n, 0.25)
    assert np.array_equal(dv.clip(X, dv.DVec.from_host(lb), dv.DVec.from_host(ub)).to_host(),
                          np.minimum(np.maximum(x, lb), ub))
    # odd-offset views take the 8-byte path
    if n > 3:
        assert np.array_equal((X[1:] + Y[1:]).to_host(), x[1:] + y[1:])
        assert np.array_equal((X[1:n - 1] - Y[2:]).to_host(), x[1:n - 1] - y[2:])


@pytest.mark.parametrize("n", [1, 63, 4097, 1 << 20])
def test_reductions(dv, n):
    rng = np.random.default_rng(n + 1)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    X, Y = dv.DVec.from_host(x), dv.DVec.from_host(y)
    assert abs(X.dot(Y) - x.dot(y)) <= 1e-13 * np.sqrt(n) * max(1, abs(x.dot(y)))
    assert abs(dv.norm(X) - np.linalg.norm(x)) <= 1e-14 * np.linalg.norm(x)
    assert dv.norm_inf(X) == np.abs(x).max()
    # determinism: same bits on every call
    assert X.dot(Y) == X.dot(Y)
    lb, ub = -np.abs(rng.standard_normal(n)), np.abs(rng.standard_normal(n))
    cnt = dv.count_outside_box(X, dv.DVec.from_host(lb), dv.DVec.from_host(ub))
    assert cnt == np.count_nonzero((x < lb) | (x > ub))


def test_scalar_pack(dv):
    """One read for a whole decision point of the SQP loop (ipsolver/sqp.py): every value is
    the bit the unpacked call returns; a one-off reduction issued between two enqueues (an
    operator forming its own norm) does not disturb the enqueued slots; two packs with unread
    slots at once are refused; empty vectors read as 0."""
    from ipsolver import _hip
    rng = np.random.default_rng(11)
    x, y = rng.standard_normal(5000), rng.standard_normal(5000)
    X, Y = dv.DVec.from_host(x), dv.DVec.from_host(y)
    E = dv.DVec.from_host(np.empty(0))
    want = [X.dot(Y), dv.norm(X), dv.norm_inf(Y), 0.0, dv.norm(Y)]
    pk = dv.ScalarPack()
    h = [pk.dot(X, Y), pk.norm(X)]
    assert dv.norm(Y) == want[4] and X.dot(X) > 0          # one-off reductions in between
    h += [pk.norm_inf(Y), pk.norm(E), pk.norm(Y)]
    other = dv.ScalarPack()
    with pytest.raises(_hip.IpxError):
        other.norm(X)
    vals = pk.read()
    assert [vals[k] for k in h] == want
    h2 = other.norm(X)                                      # the first pack was read: free again
    assert other.read()[h2] == want[1]


def test_box_sphere_reduce(dv):
    rng = np.random.default_rng(3)
    n = 50001
    z, d = rng.standard_normal(n), rng.standard_normal(n)
    d[::7] = 0.0
    lb, ub = z - np.abs(rng.standard_normal(n)), z + np.abs(rng.standard_normal(n))
    lb[::5], ub[::3] = -np.inf, np.inf
    z[14] = ub[14] + 1.0            # a d==0 coordinate outside the box
    alpha = 0.37
    out = dv.box_sphere_reduce(dv.DVec.from_host(z), dv.DVec.from_host(d), alpha,
                               dv.DVec.from_host(lb), dv.DVec.from_host(ub))
    dd = alpha * d
    nz = dd != 0
    t1, t2 = (lb[nz] - z[nz]) / dd[nz], (ub[nz] - z[nz]) / dd[nz]
    assert abs(out[0] - dd.dot(dd)) < 1e-10 and abs(out[1] - z.dot(dd)) < 1e-10
    assert abs(out[2] - z.dot(z)) < 1e-9
    assert out[3] == np.max(np.minimum(t1, t2))
    assert out[4] == np.min(np.maximum(t1, t2))
    assert out[5] == np.count_nonzero(~nz & ((z < lb) | (z > ub)))
    assert out[6] == np.count_nonzero(nz)


def _random_csr(m, n, density, rng, empty_rows=True):
    M = sps.random(m, n, density=density, format="csr", random_state=np.random.RandomState(5),
                   data_rvs=rng.standard_normal)
    if empty_rows and m > 4:
        M = sps.csr_matrix(M.toarray() * (np.arange(m)[:, None] % 4 != 1))
    M.sort_indices()
    return M


@pytest.mark.parametrize("shape", [(1, 1), (5, 9), (300, 200), (2000, 2000), (40, 5000)])
def test_csr_spmv_bitexact(dv, shape):
    rng = np.random.default_rng(11)
    m, n = shape
    M = _random_csr(m, n, min(1.0, 20.0 / n), rng)
    x = rng.standard_normal(n)
    A = dv.DeviceCSR.from_scipy(M)
    X = dv.DVec.from_host(x)
    assert np.array_equal(A.dot(X).to_host(), M.dot(x))
    y = rng.standard_normal(m)
    assert np.array_equal(A.T.dot(dv.DVec.from_host(y)).to_host(), sps.csr_matrix(M.T).dot(y))
    # fused epilogue: r - A x with sum of squares
    r = rng.standard_normal(m)
    out = A.spmv(X, alpha=-1.0, beta=1.0, yin=dv.DVec.from_host(r), reduce=True)
    want = r - M.dot(x)
    assert np.array_equal(out.to_host(), want)
    red = dv.ctx().out[:2].tolist()
    assert abs(red[0] - want.dot(want)) <= 1e-12 * max(1.0, want.dot(want))


def test_csr_spmv_long_rows_and_banded(dv):
    rng = np.random.default_rng(12)
    # one dense row far beyond the LDS tile, among short ones
    dense = np.zeros((6, 9000))
    dense[2, :] = rng.standard_normal(9000)
    dense[0, :3] = 1.0
    dense[5, 17] = -2.0
    M = sps.csr_matrix(dense)
    x = rng.standard_normal(9000)
    got = dv.DeviceCSR.from_scipy(M).dot(dv.DVec.from_host(x)).to_host()
    want = M.dot(x)
    assert np.array_equal(got[[0, 1, 3, 4, 5]], want[[0, 1, 3, 4, 5]])
    assert abs(got[2] - want[2]) <= 1e-12 * np.abs(dense[2]).dot(np.abs(x))
    # Appendix C banded Jacobian and tridiagonal Hessian with the p'Hp epilogue
    from banded_setup import BandedInstance
    inst = BandedInstance(20000, 2000)
    A = dv.DeviceCSR.from_scipy(inst.A)
    H = dv.DeviceCSR.from_scipy(inst.H)
    p = rng.standard_normal(20000)
    P = dv.DVec.from_host(p)
    assert np.array_equal(A.dot(P).to_host(), inst.A.dot(p))
    hp = H.spmv(P, reduce=True)
    red = dv.ctx().out[:2].tolist()
    assert np.array_equal(hp.to_host(), inst.H.dot(p))
    assert abs(red[1] - p.dot(inst.H.dot(p))) <= 1e-12 * abs(p.dot(inst.H.dot(p)))
    d = rng.standard_normal(20000)
    hp2 = H.spmv(P, diag=dv.DVec.from_host(d))
    assert np.array_equal(hp2.to_host(), inst.H.dot(p) + d * p)


def test_hessian_terms_merge_on_device():
    """Several sparse Lagrangian-Hessian terms (objective + constraints) become one
    CSR matrix on the union of their patterns, values summed on the device
    (backend_hip._merge_sparse_terms); diagonal terms add up as vectors.  The
    operator must act like the sum of the terms (tr_interior_point.py:222-241)."""
    import scipy.sparse as sps
    from ipsolver import backend_hip as bh
    from ipsolver.device import DVec
    rng = np.random.default_rng(5)
    n = 3000

    def sym(density, seed):
        M = sps.random(n, n, density=density, random_state=seed, format="csr")
        return sps.csr_matrix(M + M.T)
    terms = [sym(0.002, 1), sps.diags(rng.standard_normal(n), format="csr"), sym(0.001, 2),
             sps.diags([rng.standard_normal(n - 1), rng.standard_normal(n), rng.standard_normal(n - 1)],
                       [-1, 0, 1], format="csr")]
    total = sum(terms[1:], terms[0])
    p = rng.standard_normal(n)
    for slack in (None, DVec.from_host(rng.uniform(1, 2, 40))):
        H = bh.hessian_operator(list(terms), n, slack)
        assert H.csr is not None and not H.others          # one fused SpMV, usable by the CG loop
        if slack is None:
            got = H.dot(DVec.from_host(p)).to_host()
            want = total.dot(p)
        else:
            ps = rng.standard_normal(40)
            got = H.dot(DVec.from_host(np.concatenate((p, ps)))).to_host()
            want = np.concatenate((total.dot(p), slack.to_host() * ps))
        assert np.max(np.abs(got - want)) <= 1e-13 * np.max(np.abs(want))
    # same patterns, new values: the cached union pattern is reused
    terms2 = [t * 2.0 for t in terms]
    H2 = bh.hessian_operator(list(terms2), n, None)
    assert H2.csr.pattern is bh.hessian_operator(list(terms), n, None).csr.pattern
    assert np.max(np.abs(H2.dot(DVec.from_host(p)).to_host() - 2 * total.dot(p))) \
        <= 1e-13 * np.max(np.abs(total.dot(p)))


@pytest.mark.parametrize("m,n", [(3, 8), (16, 4), (77, 1001), (130, 64), (64, 33), (200, 1000),
                                 (513, 2050)])
def test_gram_mfma_tiled(m, n):
    """G = A A' on the fp64 matrix cores (csrc/dense.hip k_gram_mfma: 64 x 64 tiles through
    LDS) against numpy, at sizes that are no multiple of the tile, of the K chunk, or of the
    16-byte load (odd row length -> scalar loads); padded tail = identity."""
    import torch
    from ipsolver import _hip, device as dv
    lib = _hip.load()
    rng = np.random.default_rng(m * 1000 + n)
    A_h = rng.standard_normal((m, n))
    A = torch.from_numpy(A_h).cuda()
    M = int(lib.ipx_dense_padded(m))
    G = torch.full((M, M), np.nan, dtype=torch.float64, device="cuda")
    _hip.call("ipx_gram_f64_mfma", m, n, dv._p(A), n, dv._p(G), dv.stream_ptr())
    Gh = G.cpu().numpy()
    ref = A_h @ A_h.T
    assert np.max(np.abs(Gh[:m, :m] - ref)) <= 1e-13 * np.max(np.abs(ref))
    assert np.array_equal(Gh, Gh.T)                       # mirrored exactly
    tail = np.eye(M)
    tail[:m, :m] = Gh[:m, :m]
    assert np.array_equal(Gh, tail)                       # identity in the padding


@pytest.mark.parametrize("m", [5, 64, 130, 300, 450, 2000])
def test_dense_cholesky_and_inverse_on_the_matrix_cores(m):
    """G = L L' (64 x 64 tiles: diagonal tile in LDS, panel by substitution, trailing update on
    the fp64 matrix cores) and G^-1 (in-place triangular inverse by recursive doubling, X'X as
    MFMA tiles: csrc/dense.hip) against numpy, at tile counts that are a power of two, odd (the
    ragged last pair of every level) and one; padded tail = identity."""
    import torch
    from ipsolver import _hip, device as dv
    lib = _hip.load()
    rng = np.random.default_rng(m)
    n = 3 * m + 7
    A_h = rng.standard_normal((m, n))
    A = torch.from_numpy(A_h).cuda()
    M = int(lib.ipx_dense_padded(m))
    assert M % 64 == 0
    G = torch.empty((M, M), dtype=torch.float64, device="cuda")
    st = dv.stream_ptr()
    _hip.call("ipx_gram_f64_mfma", m, n, dv._p(A), n, dv._p(G), st)
    G_h = G.cpu().numpy()
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    work = torch.zeros(M + 1, dtype=torch.float64, device="cuda")
    _hip.call("ipx_chol_factor", M, dv._p(G), dv._p(flag), dv._p(work), st)
    assert int(flag.item()) == 0
    L = np.tril(G.cpu().numpy())
    L_ref = np.linalg.cholesky(G_h)
    assert np.max(np.abs(L - L_ref)) <= 1e-12 * np.max(np.abs(L_ref))
    d = np.diag(L_ref) ** 2 / np.diag(G_h)
    assert abs(float(work[M].item()) - d.min()) <= 1e-10 * d.min()     # min pivot / diagonal
    X = torch.full((M, M), np.nan, dtype=torch.float64, device="cuda")
    _hip.call("ipx_chol_inverse", M, dv._p(G), dv._p(X), st)
    X_h = X.cpu().numpy()
    assert np.array_equal(X_h, X_h.T)
    resid = np.max(np.abs(X_h @ G_h - np.eye(M)))
    assert resid <= 1e-9 * np.linalg.cond(G_h[:m, :m]), resid
    tail = np.eye(M)
    tail[:m, :m] = X_h[:m, :m]
    assert np.allclose(X_h, tail, rtol=0, atol=1e-300)          # identity in the padding


def test_read_back_entries(dv):
    """ipx_read_doubles / ipx_read_folded (the blocking reads of the outer loops): the values of
    the device array, bit for bit, for every count the entry accepts; folded reads equal the
    two-launch reductions bit for bit (same fold order) for sums, maxima and minima."""
    import ctypes
    import torch
    from ipsolver import _hip
    lib = _hip.load()
    rng = np.random.default_rng(5)
    for k in (1, 2, 15, 16, 17, 511, 512):
        a = rng.standard_normal(600)
        t = torch.from_numpy(a).cuda()
        assert dv.read_doubles(t, k) == list(a[:k])
        assert dv.read_doubles(t, k, 37) == list(a[37:37 + k])
    assert dv.read_doubles(torch.zeros(4, dtype=torch.float64, device="cuda"), 0) == []
    buf = (ctypes.c_double * 600)()
    assert lib.ipx_read_doubles(t.data_ptr(), 513, buf, dv.stream_ptr()) != 0      # over the limit
    # (many reads in a row: every one sees ITS values -- the tags of the pinned granules)
    for rep in range(200):
        t.fill_(float(rep))
        assert dv.read_doubles(t, 3) == [float(rep)] * 3
    # folded reads against the two-launch forms
    for n in (1, 1000, 1023, 1025, 300001, 1 << 20):
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        X, Y = dv.DVec.from_host(x), dv.DVec.from_host(y)
        c = dv.ctx()
        _hip.call("ipx_dot", n, dv._p(X.t), dv._p(Y.t), dv._p(c.out), dv._p(c.ws), dv.stream_ptr())
        two = dv.read_doubles(c.out, 1)[0]
        assert X.dot(Y) == two                                   # (DVec.dot: the folded form)
        _hip.call("ipx_norms", n, dv._p(X.t), dv._p(c.out), dv._p(c.ws), dv.stream_ptr())
        two = dv.read_doubles(c.out, 2)
        assert X.sumsq_amax() == two and two[1] == np.abs(x).max()
        g = dv._reduce_grid(n)
        assert g == int(lib.ipx_reduce_grid(n))
    # a full arena: the reductions take their two-launch forms, same values
    x, y = rng.standard_normal(4000), rng.standard_normal(4000)
    X, Y = dv.DVec.from_host(x), dv.DVec.from_host(y)
    want = (X.dot(Y), X.sumsq_amax())
    c = dv.ctx()
    c.parts_used = dv.PARTS_ARENA
    pk = dv.ScalarPack()
    h = [pk.dot(X, Y), pk.sumsq(X), pk.norm_inf(X)]
    assert (X.dot(Y), X.sumsq_amax()) == want
    vals = pk.read()
    assert [vals[k] for k in h] == [want[0], want[1][0], want[1][1]] and c.parts_used == 0
    # minimum / maximum folds of an arbitrary partial array
    p = rng.standard_normal(777)
    c = dv.ctx()
    off = c.partials(777)
    c.parts[off:off + 777] = torch.from_numpy(p).cuda()
    got = dv.read_folded([(off, 777, dv.MAX), (off, 777, dv.MIN), (off, 5, dv.SUM)])
    c.parts_used = 0
    assert got[0] == p.max() and got[1] == p.min() and abs(got[2] - p[:5].sum()) <= 1e-15 * 5


def test_aat_band_staged_and_long_rows(dv):
    """S = A A' in band storage (ipx_aat_band): the workgroup-staged join of short rows, rows
    too long for the staging buffer (the same launch joins them out of global memory), a column
    weighting, rows given in another order -- all against scipy, entry by entry."""
    import torch
    from ipsolver import _hip
    rng = np.random.default_rng(9)
    for m, width, k in ((1000, 7, 1), (700, 40, 3), (513, 3, 2), (300, 64, 2)):
        n = m * 3 + width
        rows = []
        for i in range(m):
            cols = np.sort(rng.choice(np.arange(3 * i, 3 * i + width + 3 * k), size=width,
                                      replace=False))
            cols = cols[cols < n]
            rows.append(cols)
        indptr = np.concatenate(([0], np.cumsum([len(r) for r in rows]))).astype(np.int32)
        indices = np.concatenate(rows).astype(np.int32)
        data = rng.standard_normal(len(indices))
        A = sps.csr_matrix((data, indices, indptr), shape=(m, n))
        w = rng.uniform(0.5, 2.0, n)
        for weights, perm in ((None, None), (w, None), (None, rng.permutation(m).astype(np.int32))):
            B = A if perm is None else A[perm]
            S = (B.multiply(w) if weights is not None else B).dot(B.T).toarray()
            kk = min(max(int(np.max(np.abs(np.subtract(*np.nonzero(S))))), 1), 8)
            band = torch.empty((kk + 1) * m, dtype=torch.float64, device="cuda")
            T = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()
            ip_d, ix_d, v_d = T(indptr, np.int32), T(indices, np.int32), T(data, np.float64)
            pd = T(perm, np.int32) if perm is not None else None
            wd = T(w, np.float64) if weights is not None else None
            _hip.call("ipx_aat_band_w", m, kk, dv._p(ip_d), dv._p(ix_d), dv._p(v_d), dv._p(pd),
                      dv._p(wd), dv._p(band), dv.stream_ptr())
            got = band.cpu().numpy().reshape(kk + 1, m)
            for d in range(kk + 1):
                want = np.concatenate((np.zeros(d), np.diagonal(S, -d)))
                assert np.allclose(got[d], want, rtol=1e-13, atol=1e-13), (m, width, d)
