#!/usr/bin/env python3
"""How far do the curve searches of the emissivity-growth look-up move from segment to segment?  (CPU only.)

For a few limb rays of the bench workload the look-up chain of every (channel, gas) pair is replayed in numpy
(same tables, same line of sight from the oracle's ray tracer; indices only, so ordinary double arithmetic is good
enough) and the bracket indices of get_u and get_eps on the four (p, T) corner curves are recorded.  Printed:
  * resume-at-get_eps (what ega_eps_warm does): distance of get_u's bracket from the bracket get_eps ended in on the
    previous segment, and of get_eps's bracket from get_u's on the same segment;
  * two positions per curve: distance of each search from where the SAME search ended on the previous segment.
usage: python3 tools/ega_search_stats.py [rays=12]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import common
from oracle import orc
from jurassic_hip import synth


def curves_of(rows):
    """[(p, [(T, u[], eps[]) ...]) ...] in file order, values as the fp32 the tables store."""
    levels = []
    for p, t, u, e in rows:
        if not levels or levels[-1][0] != p:
            levels.append((p, []))
        cur = levels[-1][1]
        if not cur or cur[-1][0] != t:
            cur.append((t, [], []))
        cur[-1][1].append(np.float32(u))
        cur[-1][2].append(np.float32(e))
    return [(p, [(t, np.array(u, dtype=np.float64), np.array(e, dtype=np.float64)) for t, u, e in cs]) for p, cs in levels]


def bracket(x, v):
    return int(min(max(np.searchsorted(x, v, side="right") - 1, 0), len(x) - 2))


def lip(x0, y0, x1, y1, x):
    return y0 + (x - x0) * (y1 - y0) / (x1 - x0)


def main():
    nrays = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    geom = synth.limb_geometry(nrays, scan=True, zmin=5.0, zmax=66.0)
    case = common.limb_case(geom=geom, nu=common.CTM4_NU)
    hist = {k: {} for k in ("get_u from last get_eps", "get_eps from get_u", "get_u from last get_u", "get_eps from last get_eps")}
    nlook = 0
    for ray in geom:
        los = orc.traceray(case.ctl, case.atm, ray)
        for (g, d), rows in case.rows.items():
            lv = curves_of(rows)
            pax = np.array([p for p, _ in lv])
            tau = 1.0
            last_u, last_e = {}, {}
            for ip in range(los["np"]):
                if tau < 1e-9:
                    break
                p, t, u = los["p"][ip], los["t"][ip], los["u"][g][ip]
                ipr = bracket(pax, p)
                eps = 1.0 - tau
                ecs = []
                for lev in (ipr, ipr + 1):
                    cs = lv[lev][1]
                    it = bracket(np.array([c[0] for c in cs]), t)
                    pair = []
                    for k in (it, it + 1):
                        T, uu, ee = cs[k]
                        key = (lev - ipr, k - it)      # the kernel's slot: positions carry over when a bracket of p or T moves on
                        iu = bracket(ee, eps)
                        x = lip(ee[iu], uu[iu], ee[iu + 1], uu[iu + 1], eps) + u
                        ie = bracket(uu, x)
                        pair.append(min(max(lip(uu[ie], ee[ie], uu[ie + 1], ee[ie + 1], x), 0.0), 1.0))
                        if key in last_e:
                            for name, dist in (("get_u from last get_eps", iu - last_e[key]), ("get_eps from get_u", ie - iu),
                                               ("get_u from last get_u", iu - last_u[key]), ("get_eps from last get_eps", ie - last_e[key])):
                                hist[name][dist] = hist[name].get(dist, 0) + 1
                        last_u[key], last_e[key] = iu, ie
                    ecs.append(min(max(lip(cs[it][0], pair[0], cs[it + 1][0], pair[1], t), 0.0), 1.0))
                e = min(max(lip(pax[ipr], ecs[0], pax[ipr + 1], ecs[1], p), 0.0), 1.0)
                tau = 1.0 - e                  # tau_path *= (1 - e) / tau_path
                nlook += 1
    print("%d look-ups of %d rays x %d pairs" % (nlook, nrays, len(case.rows)))
    for name, h in hist.items():
        n = sum(h.values())
        inside = h.get(0, 0) / n
        one = (h.get(1, 0) + h.get(-1, 0)) / n
        print("%-28s stays %.3f   one bracket %.3f   further %.3f   mean |distance| %.2f   (down %.3f)" % (
            name, inside, one, 1 - inside - one, sum(abs(k) * v for k, v in h.items()) / n, sum(v for k, v in h.items() if k < 0) / n))


if __name__ == "__main__":
    main()
