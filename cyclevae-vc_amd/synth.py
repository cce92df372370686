"""Deterministic synthetic inputs and weights for the CycleVAE hot path.

Counter-based generator (splitmix64 -> uniform -> Box-Muller), so that this
container (where the goldens are made from the imported reference) and the GPU
box (where /root/reference does not exist) regenerate bit-identical float32
arrays from a (seed, name) pair.  Nothing here depends on torch's RNG.

Laws follow SURVEY.md section 8(d):
  features   x = mu + sigma * n, mu_d ~ N(0,1), sigma_d ~ U(0.5,1.5), dim 0 (uv) ~ Bernoulli(0.7)
  weights    Xavier-uniform like reference src/nets/gru_vae.py:27-31 (conv fans include kernel size)
  scale_in   diag(1/sigma), -mu/sigma   (reference train_gru_cyclevae_gauss_batch.py:344-345)
  scale_out  diag(sigma_trg), mu_trg    (reference train...:346-347)
  y_in       zeros (encoder) / (0-mu_trg)/sigma_trg (decoder)   (reference train...:357-359)
"""
import hashlib

import numpy as np

SEED = 20190721
_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def _key(seed, name):
    h = hashlib.sha256(("%d:%s" % (seed, name)).encode()).digest()
    return np.uint64(int.from_bytes(h[:8], "little"))


def uniform01(name, shape, seed=SEED):
    """float64 uniforms in (0,1), element i drawn from counter i of stream (seed,name)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + _key(seed, name)
        bits = _splitmix64(_splitmix64(ctr))
    u = ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def normal(name, shape, seed=SEED):
    """float32 N(0,1) via Box-Muller on two independent uniform streams."""
    u1 = uniform01(name + "/u1", shape, seed)
    u2 = uniform01(name + "/u2", shape, seed)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def uniform(name, shape, lo, hi, seed=SEED):
    return (lo + (hi - lo) * uniform01(name, shape, seed)).astype(np.float32)


def xavier(name, shape, seed=SEED):
    """Xavier-uniform with torch's fan rule: fan_in = shape[1]*prod(shape[2:]), fan_out = shape[0]*prod(shape[2:])."""
    rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    bound = np.sqrt(6.0 / (shape[1] * rf + shape[0] * rf))
    return uniform(name, shape, -bound, bound, seed)


def feature_stats(name, dim, seed=SEED):
    """Per-dimension (mean, scale) of a synthetic feature stream."""
    mu = normal(name + "/mu", (dim,), seed)
    sigma = uniform(name + "/sigma", (dim,), 0.5, 1.5, seed)
    return mu, sigma


def features(name, B, T, mu, sigma, uv_dim0=True, seed=SEED):
    """[B,T,D] float32 features with the given per-dim stats; dim 0 is a 0/1 voicing flag."""
    D = mu.shape[0]
    x = mu[None, None, :] + sigma[None, None, :] * normal(name + "/n", (B, T, D), seed)
    if uv_dim0:
        x[:, :, 0] = (uniform01(name + "/uv", (B, T), seed) < 0.7).astype(np.float32)
    return x.astype(np.float32)


def onehot_codes(B, T, src_is_first=True):
    """One-hot speaker codes as reference src/utils/dataset.py:73-80."""
    src = np.zeros((B, T, 2), np.float32)
    trg = np.zeros((B, T, 2), np.float32)
    src[:, :, 0 if src_is_first else 1] = 1
    trg[:, :, 1 if src_is_first else 0] = 1
    return src, trg


def gru_rnn_state(name, in_dim, out_dim, hidden, scale_in=None, scale_out=None, bias_scale=0.0,
                  kernel_size=3, seed=SEED):
    """state_dict (numpy float32) of one GRU_RNN with the reference's key names and shapes.

    scale_in / scale_out: optional (mu, sigma) pairs -> frozen (un)normalisation layers.
    bias_scale > 0 draws biases from U(+-bias_scale) instead of the reference's zero init, so parity
    tests exercise every bias path.
    """
    ks = kernel_size
    c1, c2 = in_dim * ks, in_dim * ks * ks
    tot = c2 + out_dim
    sd = {}

    def bias(k, n):
        if bias_scale > 0:
            return uniform(name + "/" + k, (n,), -bias_scale, bias_scale, seed)
        return np.zeros((n,), np.float32)

    if scale_in is not None:
        mu, sg = scale_in
        sd["scale_in.weight"] = np.diag(1.0 / sg).astype(np.float32)[:, :, None]
        sd["scale_in.bias"] = (-(mu / sg)).astype(np.float32)
    sd["conv.conv.0.weight"] = xavier(name + "/conv0.w", (c1, in_dim, ks), seed)
    sd["conv.conv.0.bias"] = bias("conv0.b", c1)
    sd["conv.conv.1.weight"] = xavier(name + "/conv1.w", (c2, c1, ks), seed)
    sd["conv.conv.1.bias"] = bias("conv1.b", c2)
    sd["gru.weight_ih_l0"] = xavier(name + "/gru.wih", (3 * hidden, tot), seed)
    sd["gru.weight_hh_l0"] = xavier(name + "/gru.whh", (3 * hidden, hidden), seed)
    sd["gru.bias_ih_l0"] = bias("gru.bih", 3 * hidden)
    sd["gru.bias_hh_l0"] = bias("gru.bhh", 3 * hidden)
    sd["out_1.weight"] = xavier(name + "/out1.w", (out_dim, hidden, 1), seed)
    sd["out_1.bias"] = bias("out1.b", out_dim)
    if scale_out is not None:
        mu, sg = scale_out
        sd["scale_out.weight"] = np.diag(sg).astype(np.float32)[:, :, None]
        sd["scale_out.bias"] = mu.astype(np.float32)
    return sd


class CycleVAEProblem(object):
    """One synthetic CycleVAE workload: encoder/decoder weights, a (B,T) feature window, codes, eps."""

    def __init__(self, B, T, in_dim=54, out_dim=50, lat_dim=32, hidden=1024, n_cyc=2, bias_scale=0.0,
                 seed=SEED, tag="w"):
        self.B, self.T = B, T
        self.in_dim, self.out_dim, self.lat_dim, self.hidden, self.n_cyc = in_dim, out_dim, lat_dim, hidden, n_cyc
        self.stdim = in_dim - out_dim
        mu, sg = feature_stats(tag + "/stats", in_dim, seed)
        self.mu, self.sigma = mu, sg
        mu_t, sg_t = mu[self.stdim:], sg[self.stdim:]
        self.enc = gru_rnn_state(tag + "/enc", in_dim, 2 * lat_dim, hidden, scale_in=(mu, sg),
                                 bias_scale=bias_scale, seed=seed)
        self.dec = gru_rnn_state(tag + "/dec", lat_dim + 2, out_dim, hidden, scale_out=(mu_t, sg_t),
                                 bias_scale=bias_scale, seed=seed)
        self.x = features(tag + "/x", B, T, mu, sg, seed=seed)
        self.cvx = features(tag + "/cvx", B, T, mu[:self.stdim], sg[:self.stdim], seed=seed)
        self.code_src, self.code_trg = onehot_codes(B, T)
        self.y_in_enc = np.zeros((B, 1, 2 * lat_dim), np.float32)
        self.y_in_dec = np.broadcast_to(((0.0 - mu_t) / sg_t).astype(np.float32)[None, None, :],
                                        (B, 1, out_dim)).copy()
        self.eps = normal(tag + "/eps", (n_cyc, 3, B, T, lat_dim), seed)


def sha256_state(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    return h.hexdigest()
