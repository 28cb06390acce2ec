"""Writes functor_kat.json: VirtualboardError known answers derived by hand from
ilcc2/include/ilcc2/Optimization.h:31-107 (board 6x8 squares, g = 0.15, HuberLoss(0.1)).
Pure closed-form arithmetic written independently of oracle/ and of the HIP kernels."""
import json
import math
import os

W, H, G, DELTA = 6, 8, 0.15, 0.1


def functor(y, z, th, ty, tz, tlw, white, oob):
    yy = math.cos(th) * y - math.sin(th) * z + ty
    zz = math.sin(th) * y + math.cos(th) * z + tz
    i = (yy + W * G / 2) / G
    j = (zz + H * G / 2) / G
    if 0 < i < W and 0 < j < H:
        both_even = (math.floor(i) % 2 == 0) and (math.floor(j) % 2 == 0)
        both_odd = (math.floor(i) % 2 == 1) and (math.floor(j) % 2 == 1)
        cell_white = tlw if (both_even or both_odd) else (not tlw)
        if cell_white == white:
            r = 0.0
        else:
            fi, fj = i - math.floor(i), j - math.floor(j)
            r = min(fi, 1 - fi) + min(fj, 1 - fj)
    elif oob:
        r = min(abs(i), abs(i - W)) + min(abs(j), abs(j - H))
    else:
        r = 0.0
    s = r * r
    rho = s if s <= DELTA * DELTA else 2 * DELTA * math.sqrt(s) - DELTA * DELTA
    return i, j, r, 0.5 * rho


ROWS = [
    # y, z, theta, ty, tz, topleftWhite, laser_white, useOutofBoard
    (0.01, 0.02, 0, 0, 0, False, True, True),
    (0.01, 0.02, 0, 0, 0, False, False, True),
    (0.01, 0.02, 0, 0, 0, True, True, True),
    (-0.44, -0.59, 0, 0, 0, False, False, True),
    (-0.44, -0.59, 0, 0, 0, False, True, True),
    (0.50, 0.10, 0, 0, 0, False, True, True),
    (0.50, 0.10, 0, 0, 0, False, True, False),
    (0.50, 0.70, 0, 0, 0, False, False, True),
    (0.10, 0.20, 0.1, 0.02, -0.03, False, False, True),
    (0.10, 0.20, 0.1, 0.02, -0.03, False, True, True),
    (0.004, 0.30, 0, 0, 0, False, False, True),
    (-0.46, 0.0, 0, 0, 0, False, True, True),
    (0.0, -0.75, 0.3, 0.0, 0.0, True, False, True),
]

out = []
for row in ROWS:
    i, j, r, half_rho = functor(*row)
    out.append(dict(y=row[0], z=row[1], theta_t=[row[2], row[3], row[4]], topleft_white=row[5],
                    laser_white=row[6], use_oob=row[7], i=i, j=j, r=r, half_rho=half_rho))
json.dump(dict(board_w=W, board_h=H, grid_length=G, huber_delta=DELTA, rows=out),
          open(os.path.join(os.path.dirname(__file__), "functor_kat.json"), "w"), indent=1)
