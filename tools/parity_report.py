#!/usr/bin/env python3
"""Round 6 (review item 5d, SURVEY §8c "report max and 99.9-percentile, zero NaNs"): the parity error of every BASELINE config against
the CPU oracle, in one table. Positions: per vertex |Pg - Pr|_2 / max(|Pr|_2, 1); normals: |Ng - Nr|_2; bar 1e-4 for both.
Every config runs on the DEFAULT plan of a fresh context (the kernel the bench line of that config names) through the C ABI.

  python tools/parity_report.py [c2 c3 c4 c5 demo sparse2 c4fk]  > profiles/r6_parity.txt      (test infrastructure: uses oracle/)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import reze_engine_amd as rz  # noqa: E402
from reze_engine_amd import synth  # noqa: E402
from helpers import fk_reference, parity_errors  # noqa: E402


def row(name, kernel, ep, en, finite, extra=""):
    print("| %s | `%s` | %d | %.3e | %.3e | %.3e | %.3e | %s | %s |" % (name, kernel, len(ep), ep.max(), np.percentile(ep, 99.9), en.max(), np.percentile(en, 99.9),
                                                                    "0" if finite else "NaN / Inf!", extra), flush=True)
    return ep.max() <= 1e-4 and en.max() <= 1e-4 and finite


def single(name, V, B, M, sparse=None):
    mesh = synth.make_mesh(V, B)
    deltas, mw, sp = None, None, None
    if sparse == "demo":
        off, idx, d3, mw = synth.make_morphs_demo_shape(V, M)
        sp = (off, idx, d3)
    elif sparse == "sparse2":
        off, idx, d3, mw = synth.make_morphs_sparse(V, M, density=0.02)
        sp = (off, idx, d3)
    elif M:
        deltas, mw = synth.make_morphs_dense(V, M)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    if deltas is not None:
        c.upload_morphs_dense(deltas)
    if sp is not None:
        c.upload_morphs_sparse(*sp)
    c.set_pose(mesh["world"], mw)
    c.deform()
    pg, ng = c.read()
    kern = c.kernel_name()
    c.close()
    cores = os.cpu_count() or 1
    if sp is not None:
        pm = oracle.morph_sparse(V, sp[0], sp[1], sp[2], mw, mesh["pos"])
        pr, nr = oracle.deform(pm, mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], None, None, threads=cores)
    else:
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], deltas, mw, threads=cores)
    ep, en = parity_errors(pg, ng, pr, nr)
    return row(name, kern, ep, en, bool(np.isfinite(pg).all() and np.isfinite(ng).all()))


def crowd(name, device_fk=False):
    V, B, I = 30000, 200, 256
    mesh = synth.make_mesh(V, B)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    c.set_instances(I)
    if device_fk:
        rng = np.random.default_rng(4242)
        q = rng.normal(size=(I, B, 4)).astype(np.float32)
        q /= np.linalg.norm(q, axis=2, keepdims=True)
        c.upload_skeleton_topology(mesh["parents"], mesh["bind"])
        c.set_pose_local(q)
        # the checker's world matrices: Model.computeWorldMatrices in float64 (tests/helpers.py), rounded to f32 for the oracle's skin
        worlds = np.stack([fk_reference(mesh["parents"], mesh["bind"], q[i]).astype(np.float32) for i in range(I)])
    else:
        worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=1000 + i) for i in range(I)]).astype(np.float32)
        c.set_pose(worlds)
    c.deform()
    kern = c.kernel_name()
    eps, ens, finite = [], [], True
    cores = os.cpu_count() or 1
    for i in range(I):
        pg, ng = c.read(i)
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[i], mesh["inv_bind"], None, None, threads=cores)
        ep, en = parity_errors(pg, ng, pr, nr)
        eps.append(ep)
        ens.append(en)
        finite = finite and bool(np.isfinite(pg).all() and np.isfinite(ng).all())
    c.close()
    return row(name, kern, np.concatenate(eps), np.concatenate(ens), finite, "all %d instances" % I + ("; hierarchy solved in f32 on the GPU against float64" if device_fk else ""))


def main():
    which = sys.argv[1:] or ["c2", "c3", "c4", "c5", "demo", "sparse2", "c4fk"]
    oracle.build()
    print("# Parity of every BASELINE config against the CPU oracle (tools/parity_report.py; SURVEY §8c metric; bar 1e-4 / 1e-4)")
    print("# %s" % time.strftime("%Y-%m-%d %H:%M:%S"))
    print("| config | kernel (default plan) | vertices compared | pos max | pos p99.9 | nrm max | nrm p99.9 | non-finite | note |")
    print("|---|---|---|---|---|---|---|---|---|")
    ok = True
    if "c2" in which:
        ok &= single("C2: 30 000 verts / 200 bones / no morphs", 30000, 200, 0)
    if "c3" in which:
        ok &= single("C3: 30 000 / 200 / 64 dense morphs", 30000, 200, 64)
    if "c4" in which:
        ok &= crowd("C4: 256 instances x 30 000 / 200")
    if "c4fk" in which:
        ok &= crowd("C4 --device-fk: the same crowd from local rotations", device_fk=True)
    if "c5" in which:
        ok &= single("C5: 1 000 000 / 256 / 64 dense morphs", 1000000, 256, 64)
        ok &= single("C5 1/8 shard: 125 184 / 256 / 64", 125184, 256, 64)
        ok &= single("C5 1/4 shard: 250 112 / 256 / 64", 250112, 256, 64)
    if "demo" in which:
        ok &= single("demo-shaped: 28 842 / 349 / 60 sparse morphs on one face region", 28842, 349, 60, sparse="demo")
    if "sparse2" in which:
        ok &= single("sparse-2 %: 28 842 / 349 / 60 sparse morphs spread", 28842, 349, 60, sparse="sparse2")
    print("# %s" % ("every config inside the bar" if ok else "A CONFIG IS OUTSIDE THE BAR"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
