# coding: utf-8
"""ORACLE — test infrastructure only.  Never imported by the product path.

numpy restatement of the CAP-UDF cell extractor, reference src/render_mc.py:201-256 `extract_mesh_CAP(ndf, grad,
resolution)`:

    for every cell (i, j, k) of the (resolution-1)^3 cells, in that loop order (:207-209)
        skip if  min(ndf over the 8 corners) > 0.008                                               (:210-214)
        res[ii][jj][kk] = -ndf if dot(grad[corner 000], grad[corner]) < 0 else +ndf                 (:219-228)
        if res.min() < 0:  vertices, triangles = mcubes.marching_cubes(res, 0)                      (:230-232)
            vertices += (i, j, k); triangles += running vertex count                                (:236-245)
    v_all = v_all / (resolution - 1) * 2 - 1                                                        (:252)

THIRD-PARTY PIECE, **parity unpinned**: `mcubes.marching_cubes` is PyMCubes 0.1.4 (reference dudf.yml), absent from
this image and from /root/reference, and no reference test or fixture pins its output.  What is restated is its
published algorithm — Lorensen–Cline marching cubes on one 2x2x2 cell: a vertex on every edge whose corner values
straddle the iso-value, at the linear-interpolation point, triangles from a 256-case table — with this repo's own
table (constructed below from a face-segment rule; PyMCubes' transcribed table cannot be consulted).  Consequences:
the VERTEX SET of every cell is table-independent and is what a PyMCubes run would produce (up to its vertex order
and its `<` / `<=` convention at exact zeros); the triangle fan inside a cell may be split differently.

Conventions fixed here (and in csrc/dudf_capudf.hip): corner c = ii + 2 jj + 4 kk; corner "inside" iff res < 0;
vertices of a cell in ascending edge id; vertex = p_a + v_a / (v_a - v_b) * (p_b - p_a) in float64 from the float32
inputs; the threshold comparison is done in float64 (numpy 1.26 scalar rules of the reference environment); the
gradient dot product is accumulated left to right in float32.
"""
import numpy as np

THRESHOLD = 0.008

CORNER = np.array([[c & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)])
EDGES = [(c, c | (1 << ax)) for ax in range(3) for c in range(8) if not (c >> ax) & 1]


def _faces():
    out = []
    for ax in range(3):
        u, v = [a for a in range(3) if a != ax]
        for side in (0, 1):
            out.append([(side << ax) | (du << u) | (dv << v) for du, dv in ((0, 0), (1, 0), (1, 1), (0, 1))])
    return out


def build_table():
    """[(edge mask, [(e0, e1, e2), ...])] * 256 by the rule in the module docstring of tools/gen_mc_table.py, written
    independently of it (sets of undirected face segments -> cycles -> oriented fans); tests compare the two."""
    eid = {frozenset(e): i for i, e in enumerate(EDGES)}
    faces = _faces()
    table = []
    for idx in range(256):
        neg = [(idx >> c) & 1 for c in range(8)]
        crossed = [i for i, (a, b) in enumerate(EDGES) if neg[a] != neg[b]]
        nbr = {e: [] for e in crossed}
        for cyc in faces:
            fe = [eid[frozenset((cyc[i], cyc[(i + 1) % 4]))] for i in range(4)]
            on = [e for e in fe if e in nbr]
            if len(on) == 2:
                pairs = [tuple(on)]
            elif len(on) == 4:
                pairs = [(fe[i - 1], fe[i]) for i in range(4) if neg[cyc[i]]]
            else:
                pairs = []
            for a, b in pairs:
                nbr[a].append(b); nbr[b].append(a)
        mid = {e: (CORNER[EDGES[e][0]] + CORNER[EDGES[e][1]]) / 2.0 for e in crossed}
        out_dir = {e: (CORNER[EDGES[e][1]] - CORNER[EDGES[e][0]]) * (1 if neg[EDGES[e][0]] else -1) for e in crossed}
        todo, tris = list(crossed), []
        while todo:
            loop = [todo[0]]
            while True:
                cand = [x for x in nbr[loop[-1]] if x not in loop]
                if not cand:
                    break
                # the smaller-id neighbour first when both are free (start of a loop), to match the generator's walk
                loop.append(nbr[loop[-1]][0] if nbr[loop[-1]][0] in cand else cand[0])
            todo = [e for e in todo if e not in loop]

            def tri_ok(lp):
                for i in range(1, len(lp) - 1):
                    n = np.cross(mid[lp[i]] - mid[lp[0]], mid[lp[i + 1]] - mid[lp[0]])
                    if min(np.dot(n, out_dir[e]) for e in (lp[0], lp[i], lp[i + 1])) < -1e-12:
                        return False
                return True
            area = sum(np.cross(mid[loop[i]], mid[loop[(i + 1) % len(loop)]]) for i in range(len(loop)))
            if sum(np.dot(area, out_dir[e]) for e in loop) < 0:
                loop = loop[::-1]
            rot = next(r for r in range(len(loop)) if tri_ok(loop[r:] + loop[:r]))
            loop = loop[rot:] + loop[:rot]
            tris += [(loop[0], loop[i], loop[i + 1]) for i in range(1, len(loop) - 1)]
        table.append((sum(1 << e for e in crossed), tris))
    return table


_TABLE = None


def table():
    global _TABLE
    if _TABLE is None:
        _TABLE = build_table()
    return _TABLE


def cell_signs(ndf, grad, i, j, k):
    """res (8,) float32 in corner order c = ii + 2 jj + 4 kk — reference :216-228."""
    g0 = grad[i, j, k]
    res = np.empty(8, dtype=np.float32)
    for c in range(8):
        ii, jj, kk = CORNER[c]
        g = grad[i + ii, j + jj, k + kk]
        d = np.float32(np.float32(g0[0] * g[0]) + np.float32(g0[1] * g[1])) + np.float32(g0[2] * g[2])
        v = ndf[i + ii, j + jj, k + kk]
        res[c] = -v if d < 0 else v
    return res


def extract_mesh_CAP(ndf, grad, resolution, threshold=THRESHOLD):
    """(vertices (V,3) float64 in [-1,1]^3, triangles (T,3) int64, cells (C,3) int64 = the cells that emitted geometry,
    in emission order).  ndf (N,N,N) float32, grad (N,N,N,3) float32."""
    ndf = np.asarray(ndf, dtype=np.float32); grad = np.asarray(grad, dtype=np.float32)
    N = resolution
    tab = table()
    # candidate cells without the triple Python loop: min over the 8 corners <= threshold (float64 comparison)
    m = ndf[:-1, :-1, :-1]
    for ii, jj, kk in CORNER[1:]:
        m = np.minimum(m, ndf[ii:N - 1 + ii, jj:N - 1 + jj, kk:N - 1 + kk])
    cand = np.argwhere(~(m.astype(np.float64) > threshold))            # C order = the reference's i, j, k loop order
    v_all, t_all, cells, v_num = [], [], [], 0
    for i, j, k in cand:
        res = cell_signs(ndf, grad, i, j, k)
        if not (res.min() < 0):
            continue
        idx = sum(1 << c for c in range(8) if res[c] < 0)
        mask, tris = tab[idx]
        slot, verts = {}, []
        for e in range(12):
            if mask >> e & 1:
                a, b = EDGES[e]
                va, vb = np.float64(res[a]), np.float64(res[b])
                t = va / (va - vb)
                slot[e] = len(verts)
                verts.append(CORNER[a] + t * (CORNER[b] - CORNER[a]) + np.array([i, j, k], dtype=np.float64))
        v_all.append(np.array(verts)); t_all.append(np.array([[slot[e] for e in t] for t in tris], dtype=np.int64) + v_num)
        cells.append((i, j, k)); v_num += len(verts)
    if not v_all:
        return np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64), np.zeros((0, 3), dtype=np.int64)
    v = np.concatenate(v_all) / (resolution - 1.0) * 2.0 - 1.0
    return v, np.concatenate(t_all), np.array(cells, dtype=np.int64)
