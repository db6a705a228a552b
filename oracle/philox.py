"""Philox4x32-10 and the draws built on it, in NumPy integer arithmetic (TEST INFRASTRUCTURE).

Restates csrc/mpe_device.h's counter layout independently so that the device-side reset
(`mpe_reset`) and synthetic actions (`mpe_random_actions`) can be checked BIT-EXACTLY:
integer work, so the bar is equality.  The generator itself is pinned to the published
known-answer vectors of Random123 (Salmon et al., SC'11; `kat_vectors` philox4x32-10 rows) in
tests/test_oracle_philox.py.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
STREAM_RESET = 0x52455345
STREAM_ACTION = 0x41435449
STREAM_CHOICE = 0x43484F49   # "CHOI"
STREAM_COMM = 0x434F4D4D     # "COMM"
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over equal-shaped uint32 arrays; returns four uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(x, dtype=np.uint64) & MASK for x in (c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return [x.astype(np.uint32) for x in (c0, c1, c2, c3)]


def uniform_pm(bits, r):
    """24-bit uniform in [0,1) -> u*(2r) - r, every operation rounded to float32."""
    u = (bits >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return u * (np.float32(2.0) * np.float32(r)) - np.float32(r)


def reset_positions(seed, batch, episode, n_agents, n_landmarks, landmark_range, world_offset=0):
    """pos [B, E, 2] float32 as mpe_reset draws them for `episode`."""
    E = n_agents + n_landmarks
    b = np.arange(batch, dtype=np.uint64) + np.uint64(world_offset)
    pos = np.zeros((batch, E, 2), np.float32)
    for pair in range((E + 1) // 2):
        c0 = b & MASK
        c1 = ((b >> np.uint64(32)) ^ np.uint64((episode >> 32) & 0xFFFFFFFF)) & MASK
        c2 = np.full(batch, pair, np.uint64)
        c3 = np.full(batch, (STREAM_RESET ^ (episode & 0xFFFFFFFF)) & 0xFFFFFFFF, np.uint64)
        o = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        for half in range(2):
            e = 2 * pair + half
            if e >= E:
                break
            r = 1.0 if e < n_agents else landmark_range
            pos[:, e, 0] = uniform_pm(o[2 * half], r)
            pos[:, e, 1] = uniform_pm(o[2 * half + 1], r)
    return pos


def reset_choices(seed, batch, episode, pops, world_offset=0):
    """choice [K, B] int32 as mpe_reset draws the per-world picks (goal landmark, ...) for `episode`."""
    b = np.arange(batch, dtype=np.uint64) + np.uint64(world_offset)
    out = np.zeros((len(pops), batch), np.int32)
    for quad in range((len(pops) + 3) // 4):
        c0 = b & MASK
        c1 = ((b >> np.uint64(32)) ^ np.uint64((episode >> 32) & 0xFFFFFFFF)) & MASK
        c2 = np.full(batch, quad, np.uint64)
        c3 = np.full(batch, (STREAM_CHOICE ^ (episode & 0xFFFFFFFF)) & 0xFFFFFFFF, np.uint64)
        o = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        for k in range(4):
            i = 4 * quad + k
            if i >= len(pops):
                break
            out[i] = ((o[k].astype(np.uint64) * np.uint64(pops[i])) >> np.uint64(32)).astype(np.int32)
    return out


def action_ids(seed, batch, step, n_agents, world_offset=0):
    """ids [A, B] int32 as mpe_random_actions draws them at global step `step`."""
    b = np.arange(batch, dtype=np.uint64) + np.uint64(world_offset)
    ids = np.zeros((n_agents, batch), np.int32)
    for quad in range((n_agents + 3) // 4):
        c0 = b & MASK
        c1 = ((b >> np.uint64(32)) ^ np.uint64((step >> 32) & 0xFFFFFFFF)) & MASK
        c2 = np.full(batch, quad, np.uint64)
        c3 = np.full(batch, (STREAM_ACTION ^ (step & 0xFFFFFFFF)) & 0xFFFFFFFF, np.uint64)
        o = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        for k in range(4):
            i = 4 * quad + k
            if i >= n_agents:
                break
            ids[i] = ((o[k].astype(np.uint64) * np.uint64(5)) >> np.uint64(32)).astype(np.int32)
    return ids


def comm_ids(seed, batch, step, n_agents, dim_c, world_offset=0):
    """words [A, B] int32 in {0..dim_c-1} as mpe_random_comm draws them at global step `step` (csrc/mpe_device.h comm_draw)."""
    b = np.arange(batch, dtype=np.uint64) + np.uint64(world_offset)
    ids = np.zeros((n_agents, batch), np.int32)
    for quad in range((n_agents + 3) // 4):
        c0 = b & MASK
        c1 = ((b >> np.uint64(32)) ^ np.uint64((step >> 32) & 0xFFFFFFFF)) & MASK
        c2 = np.full(batch, quad, np.uint64)
        c3 = np.full(batch, (STREAM_COMM ^ (step & 0xFFFFFFFF)) & 0xFFFFFFFF, np.uint64)
        o = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        for k in range(4):
            i = 4 * quad + k
            if i >= n_agents:
                break
            ids[i] = ((o[k].astype(np.uint64) * np.uint64(dim_c)) >> np.uint64(32)).astype(np.int32)
    return ids


def one_hot(ids, n=5):
    return np.eye(n, dtype=np.float32)[ids]
