"""Counter-based RNG (numpy only) so weights / synthetic inputs are bit-identical on every box.

``torch.randn`` streams differ across builds and devices; the golden fixtures under
``tests/golden`` are only meaningful if the GPU box regenerates exactly the tensors the fixture
script saw (SURVEY.md section 8d "Synthetic inputs").  Each tensor gets its own stream keyed by a
stable hash of its name, each element its own counter: value = f(seed, stream, index).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _mix(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def bits(seed: int, stream: str, n: int, lane: int = 0) -> np.ndarray:
    """n uint64 words for (seed, stream); ``lane`` selects an independent substream."""
    key = np.uint64((_fnv1a(stream) ^ (seed * 0xD1342543DE82EF95) ^ (lane * 0xA0761D6478BD642F)) & 0xFFFFFFFFFFFFFFFF)
    ctr = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _mix(_mix(ctr ^ key) + key)


def uniform(seed: int, stream: str, n: int, lane: int = 0) -> np.ndarray:
    """float64 in [0, 1) with 53 random bits."""
    return (bits(seed, stream, n, lane) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, stream: str, n: int) -> np.ndarray:
    """float64 standard normal (Box-Muller on two substreams)."""
    u1 = uniform(seed, stream, n, 1)
    u2 = uniform(seed, stream, n, 2)
    return np.sqrt(-2.0 * np.log1p(-u1)) * np.cos(2.0 * np.pi * u2)


def randint(seed: int, stream: str, n: int, hi: int, lane: int = 0) -> np.ndarray:
    """int64 uniform in [0, hi)."""
    return (bits(seed, stream, n, lane) % np.uint64(hi)).astype(np.int64)
