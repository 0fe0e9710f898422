"""Randomised walk over the C ABI's state machine on the GPU: meshes, skeletons, morph sets, instance counts, tuning keys,
pose kinds, fused consumers and the launch-shape search are changed in random order, and after every pose the deformed
mesh of a random instance is compared with the CPU oracle fed with the state the walk believes the context is in.
Catches stale plans, stale buffers and flags that outlive the data they described."""
import os

import numpy as np
import pytest

from helpers import assert_hull, assert_parity, bone_morph_reference, fk_reference, sample_reference
from reze_engine_amd import synth

pytestmark = pytest.mark.gpu


class Walk:
    def __init__(self, rz, oracle, seed):
        self.rz, self.oracle, self.rng = rz, oracle, np.random.default_rng(seed)
        self.c = rz.DeformContext(0)
        self.all_variants = self.c.get_tuning("all_variants") == 1      # the tools-only build: every kernel variant is selectable
        self.mesh = None
        self.kind = None            # morph kind: None / "dense" / "sparse"
        self.I = 1
        self.topology = False
        self.edge = None
        self.aabb = False
        self.checked = 0
        self.tuning = {}
        self.log = []

    # ---- state changes -------------------------------------------------------------------------------------------
    def new_mesh(self):
        V = int(self.rng.choice([1, 3, 257, 1023, 2048, 4099, 9001]))
        B = int(self.rng.choice([1, 2, 7, 64, 300, 300, 700]))          # (300 x 17 instances, 700 x 6: poses of more than 256 KB — pulled rows, the device block ring)
        self.mesh = synth.make_mesh(V, B, seed=int(self.rng.integers(1 << 30)))
        self.c.upload_mesh(self.mesh["pos"], self.mesh["nrm"], self.mesh["joints"], self.mesh["weights"])
        self.c.upload_skeleton(self.mesh["inv_bind"])
        self.kind, self.topology, self.edge = None, False, None      # a new mesh drops morphs / edge scale; new skeleton: topology
        self.anim = None                                               # ... and the motion was flattened for the old skeleton
        self.M = 0
        self.bm = None                                                 # bone morphs name bones of the old skeleton

    def new_bone_morphs(self):
        """PMX bone morphs for device-solved poses: a random entry set (or none) over the current skeleton and morph set."""
        B, rng = len(self.mesh["parents"]), self.rng
        n = int(rng.choice([0, 1, 5, 30]))
        if n == 0:
            self.c.upload_bone_morphs([], [], [], [])
            self.bm = None
            return
        q = rng.normal(size=(n, 4)).astype(np.float32)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        self.bm = (rng.integers(0, self.M, size=n).astype(np.uint32), rng.integers(0, B, size=n).astype(np.uint32),
                   (rng.random((n, 3), dtype=np.float32) - 0.5), q)
        self.c.upload_bone_morphs(*self.bm)

    def morphed(self, q, t, w):
        if self.bm is None:
            return q, t
        o = np.argsort(self.bm[0], kind="stable")
        return bone_morph_reference(q, t, self.bm[0][o], self.bm[1][o], self.bm[2][o], self.bm[3][o], w)

    def new_morphs(self):
        self.bm = None                                                 # ... and morphs of the old set
        V = len(self.mesh["pos"])
        which = self.rng.choice(["dense", "sparse", "none"])
        if which == "none":
            self.c.upload_morphs_dense(None)
            self.kind, self.M = None, 0
            self.anim = None
            return
        M = int(self.rng.choice([1, 3, 9, 40, 140]))
        if which == "dense":
            self.deltas, _ = synth.make_morphs_dense(V, M, seed=int(self.rng.integers(1 << 30)))
            self.c.upload_morphs_dense(self.deltas)
        else:
            region = (0, max(1, V // 3)) if self.rng.random() < 0.5 else None
            self.sp = synth.make_morphs_sparse(V, M, density=min(1.0, 40.0 / V + 0.02), seed=int(self.rng.integers(1 << 30)), region=region)[:3]
            self.c.upload_morphs_sparse(*self.sp)
        self.kind, self.M = which, M
        self.anim = None            # its (empty) morph feeds were sized for the old morph set

    def new_instances(self):
        self.I = int(self.rng.choice([1, 1, 2, 5, 13, 17]))              # (17 x 300 bones: a pose of more than 256 KB — the crowd upload path)
        if self.I > 1 and (self.edge is not None or self.aabb):
            pass                                            # allowed: the generic kernel carries the consumers
        self.c.set_instances(self.I)

    def new_tuning(self):
        key = self.rng.choice(["morph_split", "unroll", "grid_cap", "geo_lds", "nontemporal", "nt_store", "fast", "out_cap", "inst_loop", "graph",
                               "inst_block", "overlap", "zero_copy", "fuse_fk", "inst_order", "inst_subsets", "pose_prefetch", "pose_pull", "fuse_fk_plain"])
        val = {"morph_split": [0, 1, 2, 4, 8], "unroll": [0, 4, 8], "grid_cap": [0, 1, 7, 64, 2048], "geo_lds": [0, 1], "nontemporal": [0, 1],
               "nt_store": [-1, 0, 1], "fast": [-1, 0, 1], "out_cap": [-1, 0, 64, 640], "inst_loop": [-1, 0, 2, 5, 8, 9, 12, 16, 33], "graph": [0, 1],
               "inst_block": [0, 256, 512, 1024], "overlap": [-1, 0, 1], "zero_copy": [-1, 0, 1], "fuse_fk": [-1, 0, 1], "inst_order": [0, 1], "inst_subsets": [-1, 0, 1], "pose_prefetch": [-1, 0, 1], "pose_pull": [-1, 0, 1], "fuse_fk_plain": [-1, 0, 1]}[key]
        v = int(self.rng.choice(val))
        if not self.all_variants and ((key == "unroll" and v == 4) or (key == "geo_lds" and v == 1) or (key == "nontemporal" and v == 0) or (key == "inst_loop" and v == 9)):
            with pytest.raises(self.rz.capi.RzError):       # the product refuses the keys of variants it does not carry ...
                self.c.set_tuning(**{key: v})
            return                                          # ... and stays as it was
        self.c.set_tuning(**{key: v})
        self.tuning[key] = v

    def toggle_consumers(self):
        V = len(self.mesh["pos"])
        if self.rng.random() < 0.5:
            self.edge = None if self.edge is not None else self.rng.random(V).astype(np.float32)
            self.c.upload_edge_scale(self.edge)
        else:
            self.aabb = not self.aabb
            self.c.enable_aabb(self.aabb)

    def put_world(self, worlds, mw):
        """World matrices: handed over (rz_set_pose) or WRITTEN IN PLACE (ABI 7: rz_map_pose / rz_commit_pose) — as whole matrices or as
        the 48-byte rows a crowd's pull kernel reads — with something else happening between the map and the commit: a tuning key, frames
        of the resident pose. A map or a commit the context has to refuse (rows for a pose its frame reads in place, the overlapped-front
        protocol, the pull switched off since the map) must leave everything as it was: the pose is then handed over whole."""
        I, rng, capi = self.I, self.rng, self.rz.capi
        whole = worlds if I > 1 else worlds[0]
        if rng.random() < 0.5:
            self.c.set_pose(whole, mw)
            return
        rows = rng.random() < 0.5
        try:
            mats, wts = self.c.map_pose(capi.POSE_ROWS12 if rows else capi.POSE_WORLD16)
        except capi.RzError:
            self.log.append('map refused')
            self.c.set_pose(whole, mw)
            return
        w4 = np.asarray(worlds, np.float32).reshape(I, -1, 4, 4)
        mats[:] = w4[..., :3].reshape(I, -1, 12) if rows else w4.reshape(I, -1, 16)
        if wts is not None and mw is not None:
            wts[:] = mw
        r = rng.random()
        if r < 0.35:
            self.new_tuning()
        elif r < 0.6:
            try:
                self.c.deform_n(int(rng.choice([1, 3])))        # frames of the RESIDENT pose while the next one is being written
                self.log.append('frames under a mapping')
            except capi.RzError:
                pass                                            # (no resident pose of this skeleton / morph set yet: refused, nothing changed)
        try:
            self.c.commit_pose()
            self.log.append('mapped rows' if rows else 'mapped')
        except capi.RzError:
            self.log.append('commit refused')
            self.c.set_pose(whole, mw)

    # ---- pose + check ----------------------------------------------------------------------------------------------
    def pose_and_check(self):
        m, B, I, rng = self.mesh, len(self.mesh["parents"]), self.I, self.rng
        mw = None
        if self.M:
            mw = rng.random((I, self.M)).astype(np.float32)
            mw[rng.random((I, self.M)) < 0.4] = 0
            if rng.random() < 0.1:
                mw[:] = 0
        sampled = rng.random() < 0.2
        local = sampled or rng.random() < 0.4
        t = None
        if sampled:
            if not self.topology:
                self.c.upload_skeleton_topology(m["parents"], m["bind"])
                self.topology = True
            if self.anim is None:
                nk = int(rng.integers(1, 5))
                kq = rng.normal(size=(B, nk, 4)).astype(np.float32)
                kq /= np.linalg.norm(kq, axis=2, keepdims=True)
                self.anim = dict(track_bone=np.arange(B, dtype=np.int32), key_off=(np.arange(B + 1) * nk).astype(np.uint32),
                                 key_frame=np.tile(np.arange(nk, dtype=np.float32) * 7, B), key_rot=kq,
                                 key_pos=(rng.random((B, nk, 3), dtype=np.float32) - 0.5), key_interp=rng.integers(0, 128, size=(B * nk, 16)).astype(np.uint8))
                if self.M:            # morph tracks: some vertex morphs keyed directly, some fed by "group" tracks, some both, some never
                    mt = int(rng.integers(1, 5))
                    mk = [np.sort(rng.choice(30, size=int(rng.integers(1, 4)), replace=False)).astype(np.float32) for _ in range(mt)]
                    feeds = [[(int(rng.integers(0, mt)), float(rng.choice([1.0, 0.5, -0.25]))) for _ in range(int(rng.integers(0, 3)))] for _ in range(self.M)]
                    self.anim.update(mkey_off=np.cumsum([0] + [len(k) for k in mk]).astype(np.uint32), mkey_frame=np.concatenate(mk),
                                     mkey_weight=rng.random(sum(len(k) for k in mk)).astype(np.float32),
                                     feed_off=np.cumsum([0] + [len(f) for f in feeds]).astype(np.uint32),
                                     feed_track=np.array([t for f in feeds for t, _ in f], np.int32), feed_ratio=np.array([r for f in feeds for _, r in f], np.float32))
                a = self.anim
                self.c.upload_animation(a["track_bone"], a["key_off"], a["key_frame"], a["key_rot"], a["key_pos"], a["key_interp"],
                                        a.get("mkey_off"), a.get("mkey_frame"), a.get("mkey_weight"), a.get("feed_off"), a.get("feed_track"), a.get("feed_ratio"))
            if self.M and rng.random() < 0.3:
                self.new_bone_morphs()
            fr = (rng.random(I) * 30 - 3).astype(np.float32)
            self.c.set_pose_sampled(fr)
            mw = None
        elif local:
            if not self.topology:
                self.c.upload_skeleton_topology(m["parents"], m["bind"])
                self.topology = True
            q = rng.normal(size=(I, B, 4)).astype(np.float32)
            q /= np.linalg.norm(q, axis=2, keepdims=True)
            t = (rng.random((I, B, 3), dtype=np.float32) - 0.5) if rng.random() < 0.5 else None
            if self.M and rng.random() < 0.3:
                self.new_bone_morphs()
            self.c.set_pose_local(q, mw, t)
        else:
            worlds = np.stack([synth.make_pose(m["parents"], m["bind"], B, seed=int(rng.integers(1 << 30))) for _ in range(I)])
            self.put_world(worlds, mw)
        if rng.random() < 0.15:
            self.c.autotune(3)
            self.log.append('autotune')
        self.c.deform()
        if rng.random() < 0.3:
            self.c.deform_n(int(rng.choice([2, 40])))      # 40: long enough for the graph replay when that key is on
            self.log.append('deform_n')
        if rng.random() < 0.1:
            self.c.time_frames(2)
            self.log.append('time_frames')                              # replaying the frame must not change it
        i = int(rng.integers(0, I))
        world = self.c.read_world(i) if local else worlds[i]
        if sampled:
            qs, ts, ws = sample_reference(self.anim, float(fr[i]), B, self.M if self.anim.get("mkey_off") is not None else 0)
            ref = fk_reference(m["parents"], m["bind"], *self.morphed(qs, ts, ws))
            assert np.abs(world - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
        elif local:
            ref = fk_reference(m["parents"], m["bind"], *self.morphed(q[i], None if t is None else t[i], np.zeros(self.M) if mw is None else mw[i]))
            assert np.abs(world - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())
        w = np.zeros(self.M, dtype=np.float32) if mw is None else mw[i]     # no weights given: all zero
        if sampled and self.M and self.anim.get("mkey_off") is not None:
            w = ws.astype(np.float32)                                        # the motion's own morph tracks
        if self.kind == "dense":
            pm = self.oracle.morph_dense(self.deltas, w, m["pos"])
        elif self.kind == "sparse":
            pm = self.oracle.morph_sparse(len(m["pos"]), self.sp[0], self.sp[1], self.sp[2], w, m["pos"])
        else:
            pm = m["pos"]
        pr, nr = self.oracle.skin(pm, m["nrm"], m["joints"], m["weights"], self.oracle.palette(world, m["inv_bind"]))
        pg, ng = self.c.read(instance=i)
        assert_parity(pg, ng, pr, nr, "walk step, V=%d B=%d M=%d(%s) I=%d local=%s" % (len(pr), B, self.M, self.kind, I, local))
        if self.edge is not None:
            assert_hull(self.c.read_hull(instance=i), self.oracle.hull(pr, nr, self.edge), "fuzz hull instance %d" % i)
        if self.aabb:
            bb = self.c.read_aabb(i)
            assert np.abs(bb[:3] - pg.min(axis=0)).max() <= 1e-5 and np.abs(bb[3:] - pg.max(axis=0)).max() <= 1e-5, \
                "aabb V=%d B=%d M=%d(%s) I=%d inst=%d local=%s tuning=%s eff(split=%d grid=%d fast=%d) log=%s bb=%s want=%s %s" % (
                    len(pr), B, self.M, self.kind, I, i, local, self.tuning, self.c.get_tuning("effective_split"), self.c.get_tuning("effective_grid"),
                    self.c.get_tuning("effective_fast"), self.log[-6:], bb, pg.min(axis=0), pg.max(axis=0))
        self.checked += 1


# REZE_FUZZ_SEEDS=200 widens the walk for a soak run; the default keeps the suite short
@pytest.mark.parametrize("seed", list(range(1, 1 + int(os.environ.get("REZE_FUZZ_SEEDS", "16")))))
def test_random_walk_over_the_abi_state_machine(rz, rzv, oracle, seed):
    """odd seeds walk the product library, even seeds the all-variants build (every kernel variant selectable)"""
    import types
    lib = rz if seed % 2 else types.SimpleNamespace(DeformContext=rzv.DeformContext, capi=rz.capi, shard=rz.shard)
    w = Walk(lib, oracle, seed)
    w.new_mesh()
    for step in range(110):
        r = w.rng.random()
        w.log.append(round(float(r), 2))
        if r < 0.10:
            w.new_mesh()
            w.I = w.I       # the instance count survives a mesh upload
        elif r < 0.25:
            w.new_morphs()
        elif r < 0.35:
            w.new_instances()
        elif r < 0.50:
            w.new_tuning()
        elif r < 0.58:
            w.toggle_consumers()
        else:
            w.pose_and_check()
    w.pose_and_check()
    assert w.checked >= 20
    print("walk %d: %d poses checked; %s" % (seed, w.checked, ", ".join("%s x %d" % (k, w.log.count(k)) for k in (
        "mapped", "mapped rows", "map refused", "commit refused", "frames under a mapping", "autotune"))))
    w.c.close()


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106])
def test_random_walk_over_sharded_contexts(rz, oracle, seed):
    """The same idea for the multi-context paths (several vertex shards on this one GPU): shard counts, morph kinds,
    peer-direct gather on and off, roots, re-uploads and pose kinds change at random; after every frame the whole mesh —
    read shard by shard, and through the root's gathered buffer when the gather is on — must match the oracle."""
    rng = np.random.default_rng(seed)
    checked = 0
    for scene in range(4):
        V = int(rng.choice([5, 1030, 4097, 20011]))
        B = int(rng.choice([3, 40, 200]))
        G = int(rng.choice([1, 2, 3, 4]))
        mesh = synth.make_mesh(V, B, seed=int(rng.integers(1 << 30)))
        kind = rng.choice(["none", "dense", "sparse"])
        M = int(rng.choice([2, 17]))
        if kind == "dense":
            deltas, _ = synth.make_morphs_dense(V, M, seed=int(rng.integers(1 << 30)))
        elif kind == "sparse":
            sp = synth.make_morphs_sparse(V, M, density=min(1.0, 30.0 / V + 0.05), seed=int(rng.integers(1 << 30)))[:3]
            deltas = synth.sparse_to_dense(V, *sp)
        else:
            deltas, M = None, 0
        ctxs = []
        for r in range(G):
            b, n, _ = rz.shard.shard_of(V, G, r)
            shard, d = rz.shard.cut_mesh(mesh, deltas, b, n)
            c = rz.DeformContext(0)
            if n > 0:
                c.upload_mesh(shard["pos"], shard["nrm"], shard["joints"], shard["weights"])
                c.upload_skeleton(mesh["inv_bind"])
                if kind == "dense":
                    c.upload_morphs_dense(d)
                elif kind == "sparse":       # cut the PMX-order entry list to the shard, indices relative to it
                    off, idx, d3 = sp
                    keep = (idx >= b) & (idx < b + n)
                    noff = np.concatenate([[0], np.cumsum([keep[off[m]:off[m + 1]].sum() for m in range(M)])]).astype(np.uint32)
                    c.upload_morphs_sparse(noff, (idx[keep] - b).astype(np.uint32), d3[keep])
            ctxs.append((c, b, n))
        live = [(c, b, n) for c, b, n in ctxs if n > 0]
        gather_root = None
        for frame in range(6):
            if len(live) == G and rng.random() < 0.5:
                gather_root = int(rng.integers(0, G))
                rz.capi.gather_direct([c for c, _, _ in ctxs], V, root=gather_root)
            world = synth.make_pose(mesh["parents"], mesh["bind"], B, seed=int(rng.integers(1 << 30)))
            mw = None
            if M:
                mw = rng.random(M).astype(np.float32)
                mw[rng.random(M) < 0.3] = 0
            tun = []
            for c, _, _ in live:
                t = dict(grid_cap=int(rng.choice([0, 3, 512])), morph_split=int(rng.choice([0, 1, 4])), out_cap=int(rng.choice([-1, 0])))
                tun.append(t)
                c.set_tuning(**t)
                c.set_pose(world, mw)
                c.deform()
            ctxt = "seed %d V=%d B=%d G=%d %s M=%d gather_root=%s tuning=%s" % (seed, V, B, G, kind, M, gather_root, tun)
            pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], world, mesh["inv_bind"], deltas, mw)
            for c, b, n in live:
                pg, ng = c.read()
                assert_parity(pg, ng, pr[b:b + n], nr[b:b + n], "scene %d frame %d shard at %d; %s" % (scene, frame, b, ctxt))
            if gather_root is not None:
                pg, ng = ctxs[gather_root][0].read_gathered()
                assert_parity(pg, ng, pr, nr, "scene %d frame %d gathered on root %d; %s" % (scene, frame, gather_root, ctxt))
            checked += 1
        order = list(range(G))
        rng.shuffle(order)                    # destroy in random order: the root may go first
        for r in order:
            ctxs[r][0].close()
    assert checked == 24


@pytest.mark.parametrize("seed", list(range(1, 11)))
def test_random_walk_over_the_host_engine_api(tmp_path, seed):
    """tests/js/engine_fuzz.js: random Engine options and random API calls through Node -> N-API -> GPU, every rendered
    frame checked inside Node against the JS oracle."""
    import json
    import os
    import shutil
    import subprocess
    from pmx_synth import write_pmx, write_vmd
    if shutil.which("node") is None:
        pytest.skip("node is not installed on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(seed)
    (tmp_path / "m.pmx").write_bytes(write_pmx(V=int(rng.choice([700, 3000, 5000])), B=int(rng.choice([12, 40])), seed=seed, max_depth=8))
    keys = []
    for b in rng.choice(12, size=6, replace=False):
        for f in sorted(rng.choice(40, size=int(rng.integers(1, 4)), replace=False)):
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            curve = bytes(rng.integers(0, 128, size=16).astype(np.uint8)) + bytes(48)
            keys.append(("bone%d" % b, int(f), tuple(q), tuple(rng.normal(size=3) * 0.3), curve))
    (tmp_path / "a.vmd").write_bytes(write_vmd(keys, [("v1", 0, 0.8), ("v1", 20, 0.1), ("v2", 6, 0.4), ("grp", 0, 0.0), ("grp", 30, 1.0)]))
    p = subprocess.run(["node", os.path.join(root, "tests", "js", "engine_fuzz.js"), str(tmp_path / "m.pmx"), str(tmp_path / "a.vmd"), str(seed)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    r = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert r["frames"] >= 5 and r["worst"] <= 1e-4
