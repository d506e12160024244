"""TEST INFRASTRUCTURE (oracle): the reference viewer's tile rule and policy-arrow geometry, restated, plus the rasterisation rule
of csrc/gu_render.hip.  Only tests/ may import this.  Pinned by tests/golden/arrows.json (tests/test_oracle_render.py), which
tests/golden/make_golden.py captured by running the reference's own method bodies without a window.

Reference: core/envs/rendering.py -- `Viewer.__init__` :119-133 (texture per cell), `Viewer.render_policy_arrows` :159-212, the agent
trail of `Viewer.render` :287-311 (pinned by tests/golden/trail.json, captured the same way).
"""
import numpy as np

TILE_DIM = 52          # ground texture width 51 + padding 1 (rendering.py:74-75)
FULL_LENGTH = 20       # arrow_base_length_full_prob (:171)
ARROW_WIDTH = 5        # (:172)
ARROW_HEIGHT = 5       # (:173)
# the build's stand-ins for the four textures, its grid line rule and arrow colour (csrc/gu_render.hip)
COLOURS = {'ground': (220, 220, 220), 'wall': (64, 64, 64), 'lava': (220, 60, 30), 'goal': (40, 180, 60)}
ARROW_COLOUR = (20, 20, 20)
AGENT_COLOUR = (40, 90, 220)


def tile_kinds(S, goals, lava, walls):
    """rendering.py:119-133: `if is_terminal_goal(i) ... elif is_lava(i) ... elif _is_wall(i) ... else ground`."""
    goals, lava, walls = set(goals), set(lava), set(walls)
    return ['goal' if s in goals else 'lava' if s in lava else 'wall' if s in walls else 'ground' for s in range(S)]


def arrow_geoms(policy, S, goals, lava, walls):
    """rendering.py:159-212 in tile coordinates (origin = the tile's bottom-left corner, y up): for every state that is neither
    terminal (:165) nor a wall (:167) and every action with probability >= 0.1 (:176-178), in action order, the head triangle and
    the shaft from the tile centre; shaft length round(p * 20), Python's round = half to even (:175)."""
    terminal, walls = set(goals) | set(lava), set(walls)
    c = TILE_DIM // 2  # center = (x_pix_loc + tile_dim / 2).astype(int), relative to the tile (:169)
    out = []
    for s in range(S):
        if s in terminal or s in walls:
            continue
        for a, p in enumerate(np.asarray(policy[s], dtype=np.float64)):
            if p < 0.1:
                continue
            L = int(round(float(p) * FULL_LENGTH))
            w, h = ARROW_WIDTH, ARROW_HEIGHT
            if a == 0:    # up (:180-186)
                end, head = (c, c + L), [(c + w, c + L), (c - w, c + L), (c, c + L + h)]
            elif a == 2:  # down (:187-192)
                end, head = (c, c - L), [(c + w, c - L), (c - w, c - L), (c, c - L - h)]
            elif a == 1:  # right (:193-198)
                end, head = (c + L, c), [(c + L, c - w), (c + L, c + w), (c + L + h, c)]
            else:         # left (:199-204)
                end, head = (c - L, c), [(c - L, c + w), (c - L, c - w), (c - L - h, c)]
            out.append(dict(state=s, head=[list(v) for v in head], start=[c, c], end=list(end)))
    return out


def rasterise(geoms, px):
    """bool[px, px] (row 0 = top) of one tile: the rule include/gu.h states for gu_render_policy_rgb.  Exact integer arithmetic on
    coordinates scaled by 2 * 52: a 52-tile coordinate v maps to 52 * px + 2 * px * (v - 26), a pixel centre i + 1/2 to 52 * (2 i + 1)."""
    mask = np.zeros((px, px), bool)
    half_width = 52 * max(1, px // 26)  # max(1, px / 26) / 2 pixels, in the scaled units
    centres = [52 * (2 * i + 1) for i in range(px)]

    def scaled(v):
        return 52 * px + 2 * px * (v - TILE_DIM // 2)

    for g in geoms:
        (x0, y0), (x1, y1) = [tuple(scaled(v) for v in pt) for pt in (g['start'], g['end'])]
        tri = [tuple(scaled(v) for v in pt) for pt in g['head']]
        (ax, ay), (bx, by), (cx, cy) = tri  # a-b is the base edge (the two vertices on the shaft's end), c the tip
        for iy in range(px):
            py = centres[px - 1 - iy]  # y up
            for ix in range(px):
                qx = centres[ix]
                on_shaft = (min(x0, x1) <= qx <= max(x0, x1) and abs(py - y0) <= half_width) if y0 == y1 else \
                           (min(y0, y1) <= py <= max(y0, y1) and abs(qx - x0) <= half_width)
                # inside the triangle: on the tip's side of the base edge (strictly) and not outside the two other edges
                def side(ux, uy, vx, vy, wx, wy):
                    return (vx - ux) * (wy - uy) - (vy - uy) * (wx - ux)
                orient = side(ax, ay, bx, by, cx, cy)
                sgn = 1 if orient > 0 else -1
                in_head = (sgn * side(ax, ay, bx, by, qx, py) > 0 and sgn * side(bx, by, cx, cy, qx, py) >= 0
                           and sgn * side(cx, cy, ax, ay, qx, py) >= 0)
                if on_shaft or in_head:
                    mask[iy, ix] = True
    return mask


def tile_frame(W, H, kinds, px, agent=None):
    """uint8[H*px, W*px, 3]: tiles in the build's palette, grid line on the top and left edge (px >= 4), agent inset square."""
    img = np.zeros((H * px, W * px, 3), np.uint8)
    for s, kind in enumerate(kinds):
        y, x = divmod(s, W)
        colour = np.array(COLOURS[kind])
        tile = np.empty((px, px, 3), np.uint8)
        tile[:] = colour
        if px >= 4:
            tile[0, :] = colour * 3 // 4
            tile[:, 0] = colour * 3 // 4
        if agent is not None and s == agent:
            lo, hi = px // 4, px - px // 4
            tile[lo:hi, lo:hi] = AGENT_COLOUR
        img[y * px:(y + 1) * px, x * px:(x + 1) * px] = tile
    return img


def policy_frame(W, H, kinds, geoms, px):
    img = tile_frame(W, H, kinds, px)
    by_state = {}
    for g in geoms:
        by_state.setdefault(g['state'], []).append(g)
    for s, gs in by_state.items():
        y, x = divmod(s, W)
        img[y * px:(y + 1) * px, x * px:(x + 1) * px][rasterise(gs, px)] = ARROW_COLOUR
    return img


# ---- the agent trail (rendering.py:287-311 over env.last_n_states, env:92-93, 182-184, 190) ----
TRAIL_ALPHA0 = 0.3      # a = 0.3 (:289)
TRAIL_DISCOUNT = 0.96   # discount = 0.96 (:290); `a *= discount` BEFORE every entry (:295), skipped ones included
TRAIL_CORNERS = ((255, 0, 0), (255, 255, 0), (0, 255, 0), (0, 0, 255))  # glColor4f at (x0, y0), (x0 + t, y0), (x0 + t, y0 + t), (x0, y0 + t), y up


def trail_quads(last_states, current):
    """[(state of the tile, alpha)] in drawing order: newest entry first, entries on the agent's current cell skipped (:296-297)."""
    a, out = TRAIL_ALPHA0, []
    for s in reversed(list(last_states)):
        a *= TRAIL_DISCOUNT
        if s == current:
            continue
        out.append((int(s), a))
    return out


def blend_trail(img, W, px, quads):
    """csrc/gu_render.hip's integer rule, in place: corner colours interpolated bilinearly at the pixel centre (doubled coordinates,
    y up), c = (c * (65536 - A) + colour * A + 32768) >> 16 per quad, A = round(alpha * 65536)."""
    two = 2 * px
    iy, ix = np.mgrid[0:px, 0:px]
    u, v = 2 * ix + 1, 2 * (px - 1 - iy) + 1
    colour = np.stack([(255 * (two - v) + px) // two, (255 * u + px) // two, (255 * (two - u) * v + two * two // 2) // (two * two)], axis=-1).astype(np.int64)
    for s, alpha in quads:
        y, x = divmod(int(s), W)
        A = int(alpha * 65536.0 + 0.5)
        tile = img[y * px:(y + 1) * px, x * px:(x + 1) * px].astype(np.int64)
        img[y * px:(y + 1) * px, x * px:(x + 1) * px] = ((tile * (65536 - A) + colour * A + 32768) >> 16).astype(np.uint8)
    return img
