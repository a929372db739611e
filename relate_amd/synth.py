"""Synthetic genotype matrices for the scale configuration (BASELINE config 4) and its scaled-down twins.

Recipe: `nchr` chromosomes with SNP counts proportional to length, positions in Mb
strictly increasing within a chromosome and restarting near 0 on the next one (so the reference's
negative-difference rule fires, src/relateHMM.cpp:396-401); allele frequency q ~ U(0.05, 0.95) per SNP;
haplotype alleles Bernoulli(q), each haplotype copying its own allele of the previous SNP with probability
0.3 (induces LD, so `choose` is non-trivial); genotype code = 1 + allele count; 0.1 % set to 0 (missing);
`n_related` planted related pairs whose IBD state is piecewise constant along the genome with exponential
segment lengths (rate a per Mb, new state drawn from k; cf. the Markov chain of R/sim_chr.R:13-51), copying
one (IBD1) or both (IBD2) haplotypes of the first individual into the second.

Random numbers come from a counter-based generator (SplitMix64 finaliser of seed/stream/index), evaluated
with torch integer ops, so the same seed gives the same matrix on CPU and on the GPU.  Data generation is
test/bench plumbing, not part of the hot path.
"""
import numpy as np
import torch

SEED = 20240601


def _mix(x):
    # SplitMix64 finaliser on int64 tensors (wrap-around arithmetic); logical shifts emulated with masks
    def lsr(v, n):
        return (v >> n) & ((1 << (64 - n)) - 1)
    x = (x ^ lsr(x, 30)) * -4658895280553007687  # 0xBF58476D1CE4E5B9
    x = (x ^ lsr(x, 27)) * -7723592293110705685  # 0x94D049BB133111EB
    return x ^ lsr(x, 31)


def uniform(seed, stream, n, device, offset=0):
    """n doubles in [0,1): element i depends only on (seed, stream, offset+i)."""
    idx = torch.arange(offset, offset + n, dtype=torch.int64, device=device)
    key = _mix(torch.tensor(seed * 1000003 + stream, dtype=torch.int64, device=device))
    bits = _mix(idx * -7046029254386353131 + key)  # 0x9E3779B97F4A7C15
    return ((bits >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / (1 << 53))


def make_dataset(N, M, nchr=22, n_related=100, seed=SEED, device="cpu", miss=0.001, copy_p=0.3, chunk=1 << 16):
    """Returns (geno uint8 N x M numpy, pos float64[M], chr int32[M], related list of (i, j, k, a))."""
    device = torch.device(device)
    # chromosome lengths ~ human autosomes (Mb), SNP counts proportional to length
    lens = np.array([248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47,
                     51], dtype=np.float64)[:nchr]
    cnt = np.floor(lens / lens.sum() * M).astype(np.int64)
    cnt[0] += M - cnt.sum()
    chr_ = np.repeat(np.arange(1, nchr + 1, dtype=np.int32), cnt)
    u = uniform(seed, 1, M, "cpu").numpy()
    pos = np.empty(M)
    start = 0
    for c in range(nchr):
        n = int(cnt[c])
        gaps = 0.2 + 1.6 * u[start:start + n]  # mean 1, strictly positive
        p = np.cumsum(gaps)
        pos[start:start + n] = p / p[-1] * lens[c]
        start += n
    q = (0.05 + 0.9 * uniform(seed, 2, M, device))  # frequency of the counted allele
    H = 2 * N
    geno = np.empty((N, M), dtype=np.uint8)
    carry = None  # last source column of every haplotype from the previous chunk
    chr_t = torch.from_numpy(chr_.astype(np.int64)).to(device)
    for s0 in range(0, M, chunk):
        s1 = min(M, s0 + chunk)
        w = s1 - s0
        base = torch.arange(H, dtype=torch.int64, device=device)[:, None] * M + torch.arange(s0, s1, dtype=torch.int64,
                                                                                            device=device)[None, :]
        key = _mix(torch.tensor(seed * 1000003 + 3, dtype=torch.int64, device=device))
        b1 = _mix(base * -7046029254386353131 + key)
        ufresh = ((b1 >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / (1 << 53))
        fresh = (ufresh < q[s0:s1][None, :]).to(torch.uint8)
        key2 = _mix(torch.tensor(seed * 1000003 + 4, dtype=torch.int64, device=device))
        b2 = _mix(base * -7046029254386353131 + key2)
        ucopy = ((b2 >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / (1 << 53))
        copy = ucopy < copy_p
        # no copying across a chromosome start
        first = torch.ones(w, dtype=torch.bool, device=device)
        first[1:] = chr_t[s0 + 1:s1] != chr_t[s0:s1 - 1]
        if s0 > 0:
            first[0] = chr_t[s0] != chr_t[s0 - 1]
        copy = copy & ~first[None, :]
        # allele = fresh allele at the last column that did not copy (prefix maximum of source indices)
        cols = torch.arange(w, dtype=torch.int64, device=device)[None, :].expand(H, w)
        src = torch.where(copy, torch.full_like(cols, -1), cols)
        src = torch.cummax(src, dim=1).values
        if carry is None:
            src = src.clamp_min(0)
            allele = torch.gather(fresh, 1, src)
        else:
            from_prev = src < 0
            allele = torch.gather(fresh, 1, src.clamp_min(0))
            allele = torch.where(from_prev, carry[:, None].expand(H, w), allele)
        carry = allele[:, -1].clone()
        g = (1 + allele[0::2] + allele[1::2]).to(torch.uint8)
        key3 = _mix(torch.tensor(seed * 1000003 + 5, dtype=torch.int64, device=device))
        b3 = _mix(base[:N] * -7046029254386353131 + key3)
        um = ((b3 >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / (1 << 53))
        g = torch.where(um < miss, torch.zeros_like(g), g)
        geno[:, s0:s1] = g.cpu().numpy()
    # planted related pairs (rewrites the second individual's genotypes from shared haplotypes is not possible
    # after genotypes are formed, so relatedness is planted at genotype level: IBD2 copies the genotype, IBD1
    # makes the pair share one allele by construction)
    related = []
    rng = np.random.default_rng(seed)
    kinds = [((0.25, 0.5, 0.25), 0.026), ((0.5, 0.5, 0.0), 0.026), ((0.0, 1.0, 0.0), 0.026)]
    for r in range(min(n_related, N // 2)):
        i, j = 2 * r, 2 * r + 1
        k, a = kinds[r % len(kinds)]
        state = np.empty(M, dtype=np.int8)
        start = 0
        for c in range(nchr):
            n = int(cnt[c])
            p = pos[start:start + n]
            cur = rng.choice(3, p=np.array(k) / np.sum(k))
            at = 0.0
            seg = np.empty(n, dtype=np.int8)
            lo = 0
            while lo < n:
                at += rng.exponential(1.0 / a)
                hi = int(np.searchsorted(p, at, side="left"))
                hi = max(hi, lo + 1) if hi <= lo else hi
                hi = min(hi, n)
                seg[lo:hi] = cur
                lo = hi
                cur = rng.choice(3, p=np.array(k) / np.sum(k))
            state[start:start + n] = seg
            start += n
        gi, gj = geno[i], geno[j].copy()
        both = (gi > 0) & (gj > 0)
        s2 = both & (state == 2)
        gj[s2] = gi[s2]
        s1 = both & (state == 1)
        # share one allele: j = one allele of i (random) + one fresh allele (j's own first allele proxy)
        ai = gi.astype(np.int16) - 1
        pick = rng.random(M) < (ai / 2.0)  # allele drawn from i's genotype
        own = rng.random(M) < q.cpu().numpy()
        gj[s1] = (1 + pick.astype(np.int16) + own.astype(np.int16)).astype(np.uint8)[s1]
        geno[j] = gj
        related.append((i, j, k, a))
    return geno, pos, chr_, related


def all_pairs(N):
    """fast_all_pairs order (src/relateHMM.cpp:3764-3765): i ascending, j > i."""
    i, j = np.triu_indices(N, 1)
    return np.stack([i, j], axis=1).astype(np.int32)
