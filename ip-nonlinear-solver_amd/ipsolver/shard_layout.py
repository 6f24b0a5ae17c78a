"""Symbolic partition of a banded Jacobian over the ranks of the row-sharded solver
(ipsolver/sharded.py): blocks of constraint rows with the variables
they bring in, one block of halo rows either side, named sub-spaces of the rows."""
import numpy as np

ROW_BLOCK = 260          # DEC_CHUNKS * (chunk + k) of csrc/banded.hip for k = 1, chunk = 64


class ShardLayout:
    """Symbolic partition of a Jacobian pattern (CSR, sorted, first column of a row
    non-decreasing: every banded Jacobian) over ``world`` ranks.  Computed identically on
    every rank from the global pattern."""

    def __init__(self, indptr, indices, shape, world, rank, row_block=ROW_BLOCK, halo_blocks=1,
                 subsets=None):
        """``subsets``: {name: sorted row indices} -- further spaces made of SOME of the rows
        (the equality / the inequality rows of a Jacobian that is banded with the two kinds
        interleaved: ipsolver/sharded_mixed.py); a rank owns the members among its own rows
        and keeps halo copies of those among its halo rows."""
        m, n = int(shape[0]), int(shape[1])
        indptr = np.asarray(indptr, dtype=np.int64)
        indices = np.asarray(indices, dtype=np.int64)
        if m == 0 or np.any(np.diff(indptr) == 0):
            raise NotImplementedError("row sharding needs a Jacobian without empty rows")
        first = indices[indptr[:-1]]
        last = np.maximum.reduceat(indices, indptr[:-1])
        if np.any(np.diff(first) < 0) or np.any(np.diff(last) < 0):
            raise NotImplementedError("row sharding needs a banded Jacobian (first and last "
                                      "column non-decreasing from row to row)")
        nb = (m + row_block - 1) // row_block
        if nb < world:
            raise ValueError("%d constraint rows give %d blocks of %d: too few for %d ranks"
                             % (m, nb, row_block, world))
        self.m, self.n, self.world, self.rank = m, n, world, rank
        self.row_block, self.halo_rows = row_block, halo_blocks * row_block
        R = [min(m, ((r * nb) // world) * row_block) for r in range(world)] + [m]
        # a variable goes with the FIRST constraint row that touches it (variables no row touches:
        # with the next one that is touched) -- the ownership of the banded solve's fused tail
        # and of the resident loop kernel (cg_fused.fuse_vown: a workgroup owns the variables
        # whose first constraint lies in its rows), so a rank's own variables are exactly its own
        # workgroups' and both kinds of partial sums cover the same entries.  Rows i and i + 1 of
        # a banded Jacobian overlap, so this is a few columns to the right of "the first column of
        # the rank's first row" (rounds 1-4).
        starts = np.full(n + 1, m, dtype=np.int64)
        np.minimum.at(starts, indices, np.repeat(np.arange(m, dtype=np.int64), np.diff(indptr)))
        first_row = np.minimum.accumulate(starts[::-1])[::-1][:n]     # (non-decreasing)
        C = [0] + [int(np.searchsorted(first_row, R[r], side="left")) for r in range(1, world)] + [n]
        self.row_cuts, self.col_cuts = R, C
        self.ranks = []
        for r in range(world):
            E0 = max(0, R[r] - self.halo_rows)
            E1 = min(m, R[r + 1] + self.halo_rows)
            x0 = 0 if r == 0 else min(C[r], int(first[E0]))
            x1 = n if r == world - 1 else max(C[r + 1], int(last[E1 - 1]) + 1)
            # an even number of local variables (one more halo column where there is room): the
            # ELL(2) form of A' the solve's tail and the resident loop kernel read takes the
            # variables in aligned pairs
            if (x1 - x0) % 2:
                if x1 < n:
                    x1 += 1
                elif x0 > 0:
                    x0 -= 1
            self.ranks.append(dict(R0=R[r], R1=R[r + 1], E0=E0, E1=E1, c0=C[r], c1=C[r + 1],
                                   x0=x0, x1=x1))
        for r in range(world):
            me = self.ranks[r]
            if r > 0:
                le = self.ranks[r - 1]
                if me["x0"] < le["c0"] or me["E0"] < le["R0"]:
                    raise ValueError("halo of rank %d reaches beyond its neighbour: fewer "
                                     "ranks or a larger problem" % r)
            if r < world - 1:
                ri = self.ranks[r + 1]
                if me["x1"] > ri["c1"] or me["E1"] > ri["R1"]:
                    raise ValueError("halo of rank %d reaches beyond its neighbour: fewer "
                                     "ranks or a larger problem" % r)
        self.me = self.ranks[rank]
        self.subsets = {k: np.asarray(v, dtype=np.int64) for k, v in (subsets or {}).items()}
        # first global entry of every rank's own part, per space (ShardVec.to_host gathers by it)
        self.cuts = {"col": C, "row": R}
        for k, idx in self.subsets.items():
            self.cuts[k] = [int(np.searchsorted(idx, r, side="left")) for r in R]

    def geom(self, kind, rank=None):
        """(global start of the local array, local length, own_lo, own_hi) for
        ``kind`` = "col" (variables) / "row" (constraints) / a named subset of the rows."""
        d = self.ranks[self.rank if rank is None else rank]
        if kind == "col":
            return d["x0"], d["x1"] - d["x0"], d["c0"] - d["x0"], d["c1"] - d["x0"]
        if kind == "row":
            return d["E0"], d["E1"] - d["E0"], d["R0"] - d["E0"], d["R1"] - d["E0"]
        idx = self.subsets[kind]
        e0, r0, r1, e1 = (int(np.searchsorted(idx, d[k], side="left")) for k in ("E0", "R0", "R1", "E1"))
        return e0, e1 - e0, r0 - e0, r1 - e0

    def sends(self, kind):
        """How many own entries the left / right neighbour keeps as its halo."""
        r = self.rank
        left = right = 0
        if r > 0:
            _, ln, _, hi = self.geom(kind, r - 1)
            left = ln - hi                  # right halo of the left neighbour
        if r < self.world - 1:
            _, _, lo, _ = self.geom(kind, r + 1)
            right = lo                      # left halo of the right neighbour
        return left, right

    def global_len(self, kind):
        if kind in self.subsets:
            return len(self.subsets[kind])
        return self.n if kind == "col" else self.m


