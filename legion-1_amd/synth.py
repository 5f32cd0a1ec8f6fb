"""Deterministic synthetic datasets in Legion's on-disk layout (host/numpy form).

The reference ships no data; its loader (GPUGraphStore.cu:254-325) reads raw
little-endian files ``edge_src`` (int64 indptr[V+1]), ``edge_dst`` (int32
indices[E]), ``features`` (f32[V*F]), ``labels`` (int32[V]) and the three seed
sets.  This module generates graphs of the ogbn-products / ogbn-papers100M /
uk-union *shapes* (legion_server.py:6-37) from closed-form integer hashes, so
that the very same graph can be produced

* here with numpy (tests, oracle), and
* on the GPU by ``lg_synth_*`` in csrc/synth.hip (full-size bench runs),

bit for bit.  Everything is integer arithmetic on splitmix64 outputs; the
feature values are exact dyadic rationals, so float rounding never enters.

Spec (``GEN_SEED = 0x1E610``):
  deg(v)      bucket b = clz24(sm64(S_DEG + v) >> 40) capped at 24 (geometric),
              deg = lo[b] + sm64(...)[low 32] % (lo[b+1] - lo[b]);  lo[] is a
              geometric ladder d0 * 1.6**b scaled so the mean hits the target.
  nbr(e)      h = sm64(S_NBR + e), a = h >> 32; 80 % (low byte < 205) skewed
              x = a^3 >> 64 (Zipf-like), else uniform x = a; r = x*V >> 32;
              dst = (r*M + C) % V with gcd(M, V) = 1 (spreads hot ids).
  feat(v, c)  ((sm64(S_FEAT + v*F + c) >> 40) * 2**-24) - 0.5   (exact in f32)
  label(v)    sm64(S_LAB + v) % classes
  seeds       id_i = (i*M2 + C2) % V; train = first n_train, then valid, test.
"""
from __future__ import annotations

import dataclasses
import math
import os

import numpy as np

GEN_SEED = 0x1E610
S_DEG = (GEN_SEED << 32) ^ 0x0DE6
S_NBR = (GEN_SEED << 32) ^ 0x0EB2
S_FEAT = (GEN_SEED << 32) ^ 0xFEA7
S_LAB = (GEN_SEED << 32) ^ 0x1AB1
NBUCKET = 24
LADDER_RATIO = 1.6

_U64 = np.uint64
_M64 = (1 << 64) - 1


def sm64(z):
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def sm64_int(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def _coprime_multiplier(V: int, start: int) -> int:
    m = start % V
    if m < 2:
        m = 2 if V > 2 else 1
    while math.gcd(m, V) != 1:
        m += 1
    return m


def degree_ladder(mean_degree: float) -> np.ndarray:
    """int32[NBUCKET+2] ladder lo[b]; expected degree == mean_degree (approx)."""
    def ladder(d0):
        lo = [max(1, int(round(d0 * LADDER_RATIO ** b))) for b in range(NBUCKET + 2)]
        for b in range(1, len(lo)):
            lo[b] = max(lo[b], lo[b - 1] + 1)
        return lo

    def mean_of(lo):
        m = 0.0
        for b in range(NBUCKET + 1):
            p = 2.0 ** -(b + 1) if b < NBUCKET else 2.0 ** -NBUCKET
            m += p * (lo[b] + (lo[b + 1] - lo[b] - 1) / 2.0)
        return m

    a, b = 0.01, 1e4
    for _ in range(80):
        mid = math.sqrt(a * b)
        if mean_of(ladder(mid)) < mean_degree:
            a = mid
        else:
            b = mid
    return np.asarray(ladder(b), dtype=np.int32)


@dataclasses.dataclass
class SynthSpec:
    """Parameters that fully determine a synthetic dataset."""
    name: str
    V: int
    F: int
    mean_degree: float
    n_train: int
    n_valid: int
    n_test: int
    classes: int = 47
    M: int = 0       # neighbour-id scrambler multiplier (coprime with V)
    C: int = 0
    M2: int = 0      # seed-set permutation multiplier
    C2: int = 0
    ladder: np.ndarray = None

    def __post_init__(self):
        if self.ladder is None:
            self.ladder = degree_ladder(self.mean_degree)
        if not self.M:
            self.M = _coprime_multiplier(self.V, 0x9E3779B1)
            self.C = 0x7F4A7C15 % self.V
        if not self.M2:
            self.M2 = _coprime_multiplier(self.V, 0x85EBCA6B)
            self.C2 = 0xC2B2AE35 % self.V


# Shapes from legion_server.py:6-37 (V, F, seed-set sizes) and E/V for the mean degree.
SHAPES = {
    "products": dict(V=2449029, F=100, mean_degree=123718280 / 2449029, n_train=196615, n_valid=39323,
                     n_test=2213091, classes=47),
    "papers100M": dict(V=111059956, F=128, mean_degree=1615685872 / 111059956, n_train=11105995,
                       n_valid=100000, n_test=100000, classes=172),
    "uk-union": dict(V=133633040, F=256, mean_degree=5507679822 / 133633040, n_train=13363304,
                     n_valid=100000, n_test=100000, classes=172),
}


def spec_for(name: str, scale: float = 1.0, F: int | None = None) -> SynthSpec:
    """A named shape, optionally shrunk (V and seed sets scaled by ``scale``)."""
    s = dict(SHAPES[name])
    if scale != 1.0:
        s["V"] = max(64, int(s["V"] * scale))
        for k in ("n_train", "n_valid", "n_test"):
            s[k] = max(1, int(s[k] * scale))
        tot = s["n_train"] + s["n_valid"] + s["n_test"]
        if tot > s["V"]:
            s["n_test"] = max(1, s["V"] - s["n_train"] - s["n_valid"])
    if F is not None:
        s["F"] = F
    return SynthSpec(name=name if scale == 1.0 else f"{name}@{scale:g}", **s)


def degrees(spec: SynthSpec, v0: int = 0, v1: int | None = None) -> np.ndarray:
    v1 = spec.V if v1 is None else v1
    v = np.arange(v0, v1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = sm64(_U64(S_DEG) + v)
    top = (h >> _U64(40)).astype(np.uint32)            # 24 uniform bits
    # leading zeros of the 24-bit field, capped at NBUCKET
    b = np.full(top.shape, NBUCKET, dtype=np.int64)
    nz = top != 0
    b[nz] = 23 - np.floor(np.log2(top[nz].astype(np.float64))).astype(np.int64)
    lo = spec.ladder.astype(np.int64)
    span = lo[b + 1] - lo[b]
    low = (h & _U64(0xFFFFFFFF)).astype(np.int64)
    return (lo[b] + low % span).astype(np.int64)


def neighbors(spec: SynthSpec, e0: int, e1: int) -> np.ndarray:
    e = np.arange(e0, e1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = sm64(_U64(S_NBR) + e)
    a = h >> _U64(32)
    skew = (h & _U64(0xFF)) < _U64(205)
    with np.errstate(over="ignore"):
        x3 = (((a * a) >> _U64(32)) * a) >> _U64(32)
        x = np.where(skew, x3, a)
        r = (x * _U64(spec.V)) >> _U64(32)
        d = (r * _U64(spec.M) + _U64(spec.C)) % _U64(spec.V)
    return d.astype(np.int32)


def features(spec: SynthSpec, ids: np.ndarray) -> np.ndarray:
    """f32[len(ids), F] rows of the synthetic feature table."""
    ids = np.asarray(ids, dtype=np.uint64).reshape(-1, 1)
    c = np.arange(spec.F, dtype=np.uint64).reshape(1, -1)
    with np.errstate(over="ignore"):
        h = sm64(_U64(S_FEAT) + ids * _U64(spec.F) + c)
    k = (h >> _U64(40)).astype(np.float32)
    return (k * np.float32(2.0 ** -24) - np.float32(0.5)).astype(np.float32)


def labels(spec: SynthSpec, ids: np.ndarray | None = None) -> np.ndarray:
    ids = np.arange(spec.V, dtype=np.uint64) if ids is None else np.asarray(ids, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = sm64(_U64(S_LAB) + ids)
    return (h % _U64(spec.classes)).astype(np.int32)


def seed_ids(spec: SynthSpec, i0: int, i1: int) -> np.ndarray:
    i = np.arange(i0, i1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return ((i * _U64(spec.M2) + _U64(spec.C2)) % _U64(spec.V)).astype(np.int32)


@dataclasses.dataclass
class Dataset:
    spec: SynthSpec
    indptr: np.ndarray      # int64[V+1]
    indices: np.ndarray     # int32[E]
    features: np.ndarray    # f32[V, F] (may be None for huge shapes)
    labels: np.ndarray      # int32[V]
    train: np.ndarray
    valid: np.ndarray
    test: np.ndarray

    @property
    def E(self) -> int:
        return int(self.indptr[-1])


def generate(spec: SynthSpec, with_features: bool = True) -> Dataset:
    deg = degrees(spec)
    indptr = np.zeros(spec.V + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    E = int(indptr[-1])
    indices = np.empty(E, dtype=np.int32)
    step = 1 << 24
    for e0 in range(0, E, step):
        indices[e0:min(E, e0 + step)] = neighbors(spec, e0, min(E, e0 + step))
    feats = None
    if with_features:
        feats = np.empty((spec.V, spec.F), dtype=np.float32)
        rstep = max(1, (1 << 22) // spec.F)
        for v0 in range(0, spec.V, rstep):
            v1 = min(spec.V, v0 + rstep)
            feats[v0:v1] = features(spec, np.arange(v0, v1))
    n1 = spec.n_train
    n2 = n1 + spec.n_valid
    n3 = n2 + spec.n_test
    return Dataset(spec, indptr, indices, feats, labels(spec), seed_ids(spec, 0, n1), seed_ids(spec, n1, n2),
                   seed_ids(spec, n2, n3))


def write_legion_files(ds: Dataset, path: str, partition_count: int | None = None) -> None:
    """Write the dataset in the reference's raw file layout (GPUGraphStore.cu:254-325)."""
    os.makedirs(path, exist_ok=True)
    ds.indptr.astype("<i8").tofile(os.path.join(path, "edge_src"))
    ds.indices.astype("<i4").tofile(os.path.join(path, "edge_dst"))
    ds.features.astype("<f4").tofile(os.path.join(path, "features"))
    ds.labels.astype("<i4").tofile(os.path.join(path, "labels"))
    ds.train.astype("<i4").tofile(os.path.join(path, "trainingset"))
    ds.valid.astype("<i4").tofile(os.path.join(path, "validationset"))
    ds.test.astype("<i4").tofile(os.path.join(path, "testingset"))
    if partition_count:
        part = (np.arange(ds.spec.V, dtype=np.int64) % partition_count).astype("<i4")
        part.tofile(os.path.join(path, f"partition_{partition_count}_bn"))


def meta_config_line(ds: Dataset, path: str, batch_size: int, cache_bytes: int, epochs: int,
                     partition_flag: int = 0) -> str:
    """The one-line ``meta_config`` the launcher writes (legion_server.py:58-59)."""
    s = ds.spec
    return "{} {} {} {} {} {} {} {} {} {} {}".format(path if path.endswith("/") else path + "/", batch_size, s.V,
                                                    ds.E, s.F, s.n_train, s.n_valid, s.n_test, cache_bytes,
                                                    epochs, partition_flag)


def minstd_pow(exp) -> np.ndarray:
    """48271**exp mod (2**31 - 1) for a uint64 array of exponents (square and multiply, vectorised)."""
    m = _U64(2147483647)
    e = np.asarray(exp, dtype=np.uint64).copy()
    r = np.ones(e.shape, dtype=np.uint64)
    b = np.full(e.shape, 48271, dtype=np.uint64)
    while e.size and e.any():
        odd = (e & _U64(1)).astype(bool)
        r = np.where(odd, (r * b) % m, r)
        b = (b * b) % m
        e >>= _U64(1)
    return r


def lp_trainingset(ds: Dataset, n_triples: int, batch_size: int, seed: int = 1, rank: int = 0, world: int = 1) -> np.ndarray:
    """Seed list for the link-prediction trainer (reference: pytorch_extension/lp_sage.py:87-90).

    lp_sage.py splits the model output of a batch into three equal thirds ``[src | pos | neg]``; the
    reference server has no edge / negative sampler, so the seed file itself must be laid out that way.
    Every batch of ``batch_size`` (a multiple of 3) seeds is ``k = batch_size/3`` source nodes, then for
    each of them one positive (a neighbour drawn with the sampler's own minstd stream: value
    ``48271**(seed + 2t + 1)`` for global triple ``t``) and one negative (uniform node id from the next
    minstd value).  Duplicates inside a batch are legal: the pipeline keeps the reference's "last
    occurrence wins" position rule.

    ``world > 1``: the list of GPU ``rank``.  The reference's ``tid % G`` split of one seed file would shred
    the thirds, so the triples are dealt by ``src % world`` (order preserved) and every GPU gets a list of
    its own whose batches keep the layout; the triples of all lists together are those of the 1-GPU list.
    csrc/synth.hip ``legion_synth_lp_seeds`` is the same rule on the GPU (bit-identical, tested).
    """
    assert batch_size % 3 == 0
    k = batch_size // 3
    t = np.arange(n_triples, dtype=np.int64)
    srcs = ds.train[t % len(ds.train)].astype(np.int64)
    if world > 1:
        sel = (srcs % world) == rank
        t, srcs = t[sel], srcs[sel]
    n = len(t)
    x1 = minstd_pow(np.uint64(seed) + _U64(2) * t.astype(np.uint64) + _U64(1))
    x2 = (x1 * _U64(48271)) % _U64(2147483647)
    lo, hi = ds.indptr[srcs], ds.indptr[srcs + 1]
    deg = hi - lo
    pick = lo + ((x1 - _U64(1)) % np.maximum(deg, 1).astype(np.uint64)).astype(np.int64)
    pos = np.where(deg > 0, ds.indices[np.minimum(pick, len(ds.indices) - 1)], srcs)
    pos = np.where(pos < 0, srcs, pos)
    neg = ((x2 - _U64(1)) % _U64(ds.spec.V)).astype(np.int64)
    n_pad = (n + k - 1) // k * k
    j = np.arange(n_pad)
    jj = np.where(j < n, j, (j // k) * k)            # pad the last batch by repeating its first triple
    out = np.empty(n_pad // k * batch_size, dtype=np.int32)
    b, i = j // k, j % k
    out[b * batch_size + i] = srcs[jj]
    out[b * batch_size + k + i] = pos[jj]
    out[b * batch_size + 2 * k + i] = neg[jj]
    return out
