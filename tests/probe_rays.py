"""Ray sets for the hit_top probes: seeded random rays plus the edge cases the reference's predicates
are sensitive to: zero direction components (1/0 = inf, 0*inf = NaN in Aabb::hit),
axis-parallel rays lying in rect planes (NaN t), origins on/inside primitives, tangent rays."""
import numpy as np


def probe_rays(case, n_random=192, seed=7):
    rs = np.random.RandomState(seed)
    if case in ("book1", "book1_list"):
        lo, hi, eye = np.array([-12, 0, -12.0]), np.array([12, 3, 12.0]), np.array([13, 2, 3.0])
    else:  # cornell-sized scenes
        lo, hi, eye = np.array([0, 0, 0.0]), np.array([555, 555, 555.0]), np.array([278, 278, -800.0])
    rays = []
    # rays from the eye towards random points of the scene volume
    for _ in range(n_random // 2):
        tgt = lo + rs.rand(3) * (hi - lo)
        rays.append(np.concatenate([eye, tgt - eye, [rs.rand()]]))
    # rays between random interior points (secondary-ray like, un-normalised)
    for _ in range(n_random // 2):
        a = lo + rs.rand(3) * (hi - lo)
        bq = lo + rs.rand(3) * (hi - lo)
        rays.append(np.concatenate([a, (bq - a) * rs.rand() * 2, [rs.rand()]]))
    c = (lo + hi) / 2
    edge = []
    for axis in range(3):
        for sgn in (1.0, -1.0):
            d = np.zeros(3)
            d[axis] = sgn
            edge.append(np.concatenate([c, d, [0.5]]))                    # two zero components
            edge.append(np.concatenate([lo + 0.25 * (hi - lo), d, [0.25]]))
            d2 = np.ones(3) * 0.5
            d2[axis] = 0.0
            edge.append(np.concatenate([c, d2 * sgn, [0.75]]))            # one zero component
            d3 = np.zeros(3)
            d3[axis] = -0.0
            d3[(axis + 1) % 3] = sgn
            edge.append(np.concatenate([c, d3, [0.1]]))                   # negative zero
    # rays starting exactly on walls / in rect planes, moving within the plane (t = 0/0 = NaN for that rect)
    for p, d in (((0.0, 100.0, 100.0), (0.0, 1.0, 0.5)), ((100.0, 0.0, 100.0), (1.0, 0.0, 0.3)),
                 ((100.0, 555.0, 100.0), (0.3, 0.0, 1.0)), ((278.0, 554.0, 279.0), (0.0, -1.0, 0.0)),
                 ((278.0, 278.0, 278.0), (0.0, 0.0, 0.0))):
        edge.append(np.array(list(p) + list(d) + [0.5]))
    # huge / tiny directions
    edge.append(np.concatenate([eye, (c - eye) * 1e30, [0.5]]))
    edge.append(np.concatenate([eye, (c - eye) * 1e-30, [0.5]]))
    return np.array(rays + edge, dtype=np.float32)
