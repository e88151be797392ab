"""TEST-SIDE probe of the MLP value heads (nothing here is product code).

`ProbedMLPHead` is a subclass of the product `MLPHeadF32` that (a) remembers the cache of its last forward, so a test can read the
pre-activations `z` of the hidden layer (which side of relu every unit took), and (b) can be told to DIFFERENTIATE the hidden layer on a GIVEN side of
relu per unit (`branch_z` = (row indices or None, pre-activations of another run of the same head on those rows)).  relu' jumps at 0; two fp32
paths — or fp32 and float64 — put a handful of the millions of units of a step on different sides, and each such unit moves one token's whole
backward signal, so gradient comparisons are made on the SAME piecewise-linear branch while the VALUES are compared on each side's own branches.
"""
from lmrl_gym_amd.train.gpt2_f32 import MLPHeadF32


class ProbedMLPHead(MLPHeadF32):
    branch_z = None
    branch_flips = 0
    last_cache = None

    def hidden(self, x, rows):
        a, z = super().hidden(x, rows)
        if self.branch_z is None:
            return a, z
        rows_idx, zb = self.branch_z
        zz = z.clone()
        if rows_idx is None:
            zz.copy_(zb)
        else:
            zz[rows_idx.long()] = zb
        self.branch_flips = int(((zz > 0) != (z > 0)).sum())
        # the VALUES stay on this run's own branches (a = relu(z) as computed); only relu' in the backward (which reads the cached z) takes the
        # given side for the `branch_flips` near-zero units
        return a, zz

    def forward(self, x, rows):
        y, cache = super().forward(x, rows)
        self.last_cache = cache
        return y, cache

    def forward_ce(self, x, rows, targets):
        out = super().forward_ce(x, rows, targets)
        self.last_cache = out[-1]
        return out
