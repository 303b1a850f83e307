#!/usr/bin/env python
# coding: utf-8
"""Generates diffudf_amd/csrc/dudf_mc_table.h: the 256-case marching-cubes table of the CAP-UDF cell extractor
(reference src/render_mc.py:231 calls `mcubes.marching_cubes(res, 0)` on one 2x2x2 cell; PyMCubes 0.1.4 is absent from
this image and from /root/reference, so its table cannot be consulted — parity with it is UNPINNED).

The table is CONSTRUCTED, not transcribed:
  * corner c = ii + 2 jj + 4 kk of the cell (ii, jj, kk = offsets along the three grid axes), bit c of the case index
    is set when the corner value is NEGATIVE ("inside");
  * an edge carries a vertex iff its two corners differ; on every cube face the crossed edges are joined by segments —
    one segment for two crossings, and for the ambiguous face (four crossings, inside corners on a diagonal) two
    segments that each cut off ONE inside corner (a rule that depends on the face's corners only, hence crack-free
    between neighbouring cells);
  * the segments close into loops; every loop is triangulated as a fan and oriented so that the normal points from
    the inside (negative) corners to the outside.
Vertices of a cell are emitted in ascending edge order; at most 12 vertices and 5 fan... (see NT_MAX printed below)
triangles per cell.

    python tools/gen_mc_table.py            # rewrites the header
"""
import os

CORNER = [(c & 1, (c >> 1) & 1, (c >> 2) & 1) for c in range(8)]
# 12 edges: (corner a, corner b) with a < b differing in exactly one axis; ordered by axis, then by the other two bits
EDGES = []
for axis in range(3):
    for c in range(8):
        if not (c >> axis) & 1:
            EDGES.append((c, c | (1 << axis)))
EDGE_ID = {e: i for i, e in enumerate(EDGES)}
# 6 faces: axis, side -> the 4 corners in cyclic order
FACES = []
for axis in range(3):
    u, v = [a for a in range(3) if a != axis]
    for side in (0, 1):
        cyc = []
        for du, dv in ((0, 0), (1, 0), (1, 1), (0, 1)):
            cyc.append((side << axis) | (du << u) | (dv << v))
        FACES.append(cyc)


def edge_of(a, b):
    return EDGE_ID[(min(a, b), max(a, b))]


def build_case(idx):
    inside = [(idx >> c) & 1 for c in range(8)]
    crossed = [i for i, (a, b) in enumerate(EDGES) if inside[a] != inside[b]]
    if not crossed:
        return 0, []
    link = {e: [] for e in crossed}
    for cyc in FACES:
        fe = [edge_of(cyc[i], cyc[(i + 1) % 4]) for i in range(4)]          # edge i joins corners i, i+1
        fx = [i for i in range(4) if inside[cyc[i]] != inside[cyc[(i + 1) % 4]]]
        if len(fx) == 2:
            segs = [(fe[fx[0]], fe[fx[1]])]
        elif len(fx) == 4:                                                     # ambiguous: cut off each inside corner
            segs = []
            for i in range(4):
                if inside[cyc[i]]:
                    segs.append((fe[(i - 1) % 4], fe[i]))                    # the two face edges meeting at corner i
        else:
            segs = []
        for a, b in segs:
            link[a].append(b); link[b].append(a)
    assert all(len(v) == 2 for v in link.values()), (idx, link)
    mid = {e: [(CORNER[EDGES[e][0]][k] + CORNER[EDGES[e][1]][k]) / 2.0 for k in range(3)] for e in crossed}
    tris, seen = [], set()
    for start in crossed:
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        while True:
            seen.add(cur)
            nxt = link[cur][0] if link[cur][0] != prev else link[cur][1]
            if len(loop) > 1 and nxt == start:
                break
            if nxt == start:
                break
            loop.append(nxt); prev, cur = cur, nxt
        # orientation: Newell normal against "inside -> outside" along the loop's own edges
        n = [0.0, 0.0, 0.0]
        for i in range(len(loop)):
            p, q = mid[loop[i]], mid[loop[(i + 1) % len(loop)]]
            n[0] += (p[1] - q[1]) * (p[2] + q[2]); n[1] += (p[2] - q[2]) * (p[0] + q[0]); n[2] += (p[0] - q[0]) * (p[1] + q[1])
        s = 0.0
        for e in loop:
            a, b = EDGES[e]
            pin, pout = (CORNER[a], CORNER[b]) if inside[a] else (CORNER[b], CORNER[a])
            s += sum(n[k] * (pout[k] - pin[k]) for k in range(3))
        assert abs(s) > 1e-9, idx
        if s < 0:
            loop = loop[::-1]
        # fan pivot: the rotation whose triangles (edge-midpoint geometry) all face outward along their own edges
        def violations(lp):
            v = 0
            for i in range(1, len(lp) - 1):
                P = [mid[lp[0]], mid[lp[i]], mid[lp[i + 1]]]
                a = [P[1][k] - P[0][k] for k in range(3)]; b = [P[2][k] - P[0][k] for k in range(3)]
                nn = [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
                for e in (lp[0], lp[i], lp[i + 1]):
                    ca, cb = EDGES[e]
                    pin, pout = (CORNER[ca], CORNER[cb]) if inside[ca] else (CORNER[cb], CORNER[ca])
                    if sum(nn[k] * (pout[k] - pin[k]) for k in range(3)) < -1e-12:
                        v += 1
            return v
        rots = [loop[r:] + loop[:r] for r in range(len(loop))]
        loop = min(rots, key=lambda lp: (violations(lp), rots.index(lp)))
        assert violations(loop) == 0, (idx, loop)
        for i in range(1, len(loop) - 1):
            tris.append((loop[0], loop[i], loop[i + 1]))
    mask = 0
    for e in crossed:
        mask |= 1 << e
    return mask, tris


def build_table():
    return [build_case(i) for i in range(256)]


def main():
    tab = build_table()
    nt_max = max(len(t) for _, t in tab)
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(here), "diffudf_amd", "csrc", "dudf_mc_table.h")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_mc_table.py — do not edit.  256-case marching-cubes table of the CAP-UDF cell extractor\n"
                "// (constructed from the face-segment rule described in that script; PyMCubes' own table is not available here).\n"
                "// corner c = ii + 2 jj + 4 kk; bit c of the case index = corner value negative; edges: see kMcEdgeCorner.\n"
                "#pragma once\n#include <stdint.h>\n#ifndef DUDF_MC_QUAL\n#define DUDF_MC_QUAL static const   /* device code defines it as __constant__ */\n#endif\n")
        f.write(f"#define DUDF_MC_MAX_TRI {nt_max}\n")
        f.write("DUDF_MC_QUAL uint8_t kMcEdgeCorner[12][2] = {" + ", ".join("{%d, %d}" % e for e in EDGES) + "};\n")
        f.write("// per case: crossed-edge mask (12 bits)\nDUDF_MC_QUAL uint16_t kMcEdgeMask[256] = {\n")
        for r in range(0, 256, 16):
            f.write("    " + ", ".join("0x%03x" % tab[i][0] for i in range(r, r + 16)) + ",\n")
        f.write("};\n// per case: triangle count, then up to DUDF_MC_MAX_TRI triangles as edge ids (0xff = unused)\n")
        f.write(f"DUDF_MC_QUAL uint8_t kMcTri[256][{1 + 3 * nt_max}] = {{\n")
        for i in range(256):
            row = [len(tab[i][1])]
            for t in tab[i][1]:
                row += list(t)
            row += [255] * (1 + 3 * nt_max - len(row))
            f.write("    {" + ", ".join(str(v) for v in row) + "},\n")
        f.write("};\n")
    print("wrote", out, "max triangles per case:", nt_max, "total triangles:", sum(len(t) for _, t in tab))


if __name__ == "__main__":
    main()
