"""The map update's walk keeps Bresenham's error term in fixed point and reads the step across the ray off the add's carry
(csrc/rbpf_raycast.hip, `advance`).  This restates that arithmetic in numpy — for every ray shape the kernel can meet
(dmaj <= kBoxSideMax = 176, dmin <= dmaj), every segment start, and a reciprocal one ulp either side of the correctly rounded one
(v_rcp_f32 is good to one ulp) — and holds the sequence of steps against the integer recurrence the reference's lineLow / lineHigh
perform (grid_mapper.cpp:707-807: `D > 0`, restated as rem + 2 dmin > 2 dmaj)."""
import numpy as np

K_BOX_SIDE_MAX = 176


def _integer_sides(dmaj, dmin, n0):
    """side[n] for n = n0 .. dmaj - 1, and the error term at n0, by the integer rule."""
    a0 = 2 * dmin * n0 - dmaj
    c0 = -(-a0 // (2 * dmaj)) if a0 > 0 else 0
    rem = a0 - 2 * dmaj * (c0 - 1)
    rem0 = rem
    out = []
    for _ in range(n0, dmaj):
        r2 = rem + 2 * dmin
        side = r2 > 2 * dmaj
        rem = r2 - 2 * dmaj if side else r2
        out.append(side)
    return rem0, np.array(out, dtype=bool)


def test_fixed_point_carry_is_the_integer_step_for_every_ray_shape():
    scale = np.float32(4294963200.0)  # 2^32 (1 - 2^-20)
    checked = 0
    for dmaj in range(1, K_BOX_SIDE_MAX + 1):
        rcp = np.float32(1.0) / np.float32(dmaj)
        for ulp in (-1, 0, 1):
            r = np.nextafter(rcp, np.float32(2.0 if ulp > 0 else 0.0)) if ulp else rcp
            R = int(np.float32(r * scale))      # (float -> u32 truncates; the product is rounded to float first, as on the device)
            assert R * dmaj < 2 ** 32
            for dmin in range(0, dmaj + 1):
                F = dmin * R
                assert F < 2 ** 32
                for n0 in sorted({0, 1, dmaj // 3, dmaj // 2, dmaj - 1}):
                    if n0 >= dmaj:
                        continue
                    rem0, want = _integer_sides(dmaj, dmin, n0)
                    assert 1 <= rem0 <= 2 * dmaj
                    acc = rem0 * (R >> 1) - 1
                    assert 0 <= acc < 2 ** 32
                    got = np.empty(len(want), dtype=bool)
                    for i in range(len(want)):
                        nxt = acc + F
                        got[i] = nxt >= 2 ** 32
                        acc = nxt & 0xFFFFFFFF
                    assert np.array_equal(got, want), (dmaj, dmin, n0, ulp)
                    checked += 1
    assert checked > 100000


def test_uniform_divisor_magic_is_the_integer_quotient_for_every_operand():
    """udiv16 (csrc/rbpf_device.hpp, round 6): n // d as the high word of (n << 8) * ceil(2^24 / d) — for every 0 <= n < 2^16 and every
    workgroup-uniform divisor the map update has (1 <= d <= 2^8: pairs in a row of the box <= 88, map tiles under it <= 8, segments per
    ray <= 4)."""
    n = np.arange(1 << 16, dtype=np.uint64)
    for d in range(1, 257):
        m = np.uint64(((1 << 24) + d - 1) // d)
        assert m < (1 << 32)
        q = ((n << np.uint64(8)) * m) >> np.uint64(32)
        assert np.array_equal(q, n // np.uint64(d)), d
