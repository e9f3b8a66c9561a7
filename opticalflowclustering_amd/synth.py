"""Deterministic synthetic inputs: an analytic multi-sinusoid
texture that can be sampled at sub-pixel shifts, so frame pairs with a known motion field can be
generated at any resolution without a video decoder.  numpy only; used by tests and bench.py."""
import numpy as np


def texture_params(seed=0, n_waves=24):
    rng = np.random.default_rng(seed)
    fx = rng.uniform(0.01, 0.12, n_waves) * rng.choice([-1.0, 1.0], n_waves)
    fy = rng.uniform(0.01, 0.12, n_waves)
    a = rng.uniform(0.3, 1.0, n_waves)
    ph = rng.uniform(0, 2 * np.pi, n_waves)
    return fx, fy, a, ph


def texture(X, Y, params):
    """T(x,y) in [0,255] (float64) at float coordinates X, Y (broadcastable arrays)"""
    fx, fy, a, ph = params
    acc = np.zeros(np.broadcast(X, Y).shape, np.float64)
    for i in range(len(a)):
        acc += a[i] * np.sin(2 * np.pi * (fx[i] * X + fy[i] * Y) + ph[i])
    return np.clip(127.5 + 100.0 * acc / np.sum(a) * 2.5, 0, 255)


def frame(W, H, dx=0.0, dy=0.0, params=None, seed=0):
    """uint8 HxW frame = floor(T(x-dx, y-dy)); dx, dy scalars or HxW arrays (content moves by +d)"""
    params = params or texture_params(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    return np.floor(texture(xx - dx, yy - dy, params)).astype(np.uint8)


def translated_pair(W, H, dx, dy, seed=0):
    p = texture_params(seed)
    return frame(W, H, 0, 0, p), frame(W, H, dx, dy, p)


def nonrigid_pair(W, H, seed=0, amp=3.0):
    """smoothly varying motion field; returns prev, next, (dx, dy)"""
    p = texture_params(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    dx = amp * np.sin(2 * np.pi * yy / H) * np.cos(np.pi * xx / W)
    dy = 0.5 * amp * np.cos(2 * np.pi * xx / W)
    return frame(W, H, 0, 0, p), frame(W, H, dx, dy, p), (dx, dy)


def noise_pair(W, H, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (H, W), dtype=np.uint8)
    b = np.roll(a, (1, 2), (0, 1))
    return a, b


def population_motion(W, H, t, n_pop=5, seed=0):
    """cfg2: per-frame piecewise motion made of n_pop motion populations (vertical bands with
    distinct velocities) so that k-means with k=n_pop over (u,v) is meaningful."""
    rng = np.random.default_rng(seed + 1000)
    vel = rng.uniform(-4, 4, (n_pop, 2))
    band = (np.arange(W) * n_pop // W)
    dx = np.broadcast_to(vel[band, 0] * (t + 1), (H, W))
    dy = np.broadcast_to(vel[band, 1] * (t + 1), (H, W))
    return dx, dy, vel
