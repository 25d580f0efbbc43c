"""MT19937 restatement of the negative-index sampler (oracle, test infrastructure only).

The reference draws its negatives with two ``torch.randint`` calls on the CPU
generator (/root/reference/cpc/criterion/criterion.py:247-256).  On the CPU
device torch's generator is the standard 32-bit Mersenne Twister and
``randint(low, high, (n,))`` consumes exactly one 32-bit output per element, in
element order, mapped as ``out % (high - low) + low``.  This file restates that
law in numpy so that indices can be checked bit-for-bit without torch.

The index recipe follows criterion.py:259-266:
    flat i = (bb * Nneg + nn) * W + t
    seq    = (seqIdx[i] + t) mod T
    extIdx = seq + batchIdx[i] * T
"""
import struct

import numpy as np

N, M = 624, 397
UPPER, LOWER, MAG = 0x80000000, 0x7FFFFFFF, 0x9908B0DF


class MT19937:
    """Mersenne Twister with torch's bookkeeping (``left`` / ``next``)."""

    def __init__(self, seed=5489):
        self.seed(seed)

    def seed(self, seed):
        mt = np.empty(N, dtype=np.uint64)
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, N):
            prev = int(mt[i - 1])
            mt[i] = (1812433253 * (prev ^ (prev >> 30)) + i) & 0xFFFFFFFF
        self.mt = mt.astype(np.uint32)
        self.left = 1   # torch: first draw triggers a twist
        self.next = 0
        self.initial_seed = seed

    # -- interop with torch.get_rng_state() / set_rng_state() (legacy CPU layout:
    #    u64 seed, i32 left, i32 seeded, u64 next, u64 state[624], 3 doubles, i32) --
    @classmethod
    def from_torch_state(cls, state_bytes):
        b = bytes(state_bytes)
        seed, left, _seeded, nxt = struct.unpack_from("<QiiQ", b, 0)
        self = cls.__new__(cls)
        self.initial_seed = seed
        self.left, self.next = left, nxt
        self.mt = np.frombuffer(b, dtype="<u8", count=N, offset=24).astype(np.uint32)
        return self

    def to_torch_state(self, template_bytes):
        b = bytearray(bytes(template_bytes))
        struct.pack_into("<QiiQ", b, 0, self.initial_seed, self.left, 1, self.next)
        b[24:24 + 8 * N] = self.mt.astype("<u8").tobytes()
        return bytes(b)

    def _twist(self):
        mt = self.mt
        new = np.empty_like(mt)

        def mix(cur, nxt, far):
            y = (cur & np.uint32(UPPER)) | (nxt & np.uint32(LOWER))
            return far ^ (y >> np.uint32(1)) ^ np.where(y & np.uint32(1), np.uint32(MAG), np.uint32(0))

        # i in [0, 227): partner i+397 is still old
        new[0:N - M] = mix(mt[0:N - M], mt[1:N - M + 1], mt[M:N])
        # i in [227, 454): partner i-227 is new[0:227]
        new[N - M:2 * (N - M)] = mix(mt[N - M:2 * (N - M)], mt[N - M + 1:2 * (N - M) + 1], new[0:N - M])
        # i in [454, 623): partner new[227:396]
        new[2 * (N - M):N - 1] = mix(mt[2 * (N - M):N - 1], mt[2 * (N - M) + 1:N], new[N - M:M - 1])
        # i = 623: wraps to new[0], partner new[396]
        new[N - 1:N] = mix(mt[N - 1:N], new[0:1], new[M - 1:M])
        self.mt = new

    @staticmethod
    def _temper(y):
        y = y ^ (y >> np.uint32(11))
        y = y ^ ((y << np.uint32(7)) & np.uint32(0x9D2C5680))
        y = y ^ ((y << np.uint32(15)) & np.uint32(0xEFC60000))
        y = y ^ (y >> np.uint32(18))
        return y

    def draw(self, n):
        """n raw 32-bit outputs, same stream/bookkeeping as torch's CPU generator."""
        out = np.empty(n, dtype=np.uint32)
        pos = 0
        while pos < n:
            # torch: if (--left == 0) twist(), left = 624, next = 0; y = state[next++]
            if self.left == 1:
                self._twist()
                self.left, self.next = N + 1, 0   # +1: the decrement below
            avail = self.left - 1                  # draws before the next twist
            take = min(avail, n - pos)
            out[pos:pos + take] = self._temper(self.mt[self.next:self.next + take])
            self.next += take
            self.left -= take
            pos += take
        return out

    def randint(self, low, high, n):
        """torch.randint(low, high, (n,)) on the CPU generator (range < 2**32)."""
        rng = np.uint32(high - low)
        return (self.draw(n) % rng).astype(np.int64) + low


def negative_indices(mt, batch, seq_len, window, n_neg):
    """Returns (batchIdx, seqIdx_raw, extIdx) int64 arrays of length n_neg*window*batch.

    Draw order: batchIdx first, then seqIdx (criterion.py:247-256).
    """
    n = n_neg * window * batch
    batch_idx = mt.randint(0, batch, n)
    seq_raw = mt.randint(1, seq_len, n)
    t = np.tile(np.arange(window, dtype=np.int64), batch * n_neg)
    seq = (seq_raw + t) % seq_len
    ext = seq + batch_idx * seq_len
    return batch_idx, seq_raw, ext
