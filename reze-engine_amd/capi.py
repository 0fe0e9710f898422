"""ctypes binding of libreze_deform.so (include/reze_deform.h) for bench.py and the GPU tests.

This is plumbing only: every method is a 1:1 call of a C-ABI entry point, so the parity tests
exercise exactly what the N-API addon binds. There is deliberately no CPU fallback — if the HIP
library is missing or no MI355X is visible, construction raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libreze_deform.so")

# every symbol include/reze_deform.h declares (tests check the library exports all of them)
SYMBOLS = [
    "rz_last_error", "rz_abi_version", "rz_device_count", "rz_device_numa_node", "rz_create", "rz_destroy", "rz_shard_range", "rz_gather_chunk",
    "rz_upload_mesh", "rz_upload_mesh_soa", "rz_upload_skeleton", "rz_upload_morphs_dense",
    "rz_upload_morphs_sparse", "rz_set_instances", "rz_set_pose", "rz_upload_skeleton_topology", "rz_set_pose_local", "rz_upload_bone_morphs", "rz_upload_animation", "rz_set_pose_sampled", "rz_override_world", "rz_read_world", "rz_deform", "rz_deform_n", "rz_fork", "rz_deform_pair", "rz_sync", "rz_read",
    "rz_read_palette", "rz_time_frames", "rz_set_tuning", "rz_get_tuning", "rz_autotune", "rz_autotune_measure", "rz_autotune_pick",
    "rz_autotune_apply", "rz_output_ptrs",
    "rz_comm_unique_id", "rz_rccl_info", "rz_comm_info", "rz_comm_init", "rz_allgather", "rz_read_gathered", "rz_comm_init_all", "rz_allgather_all", "rz_gather_direct", "rz_gather_fence", "rz_upload_edge_scale", "rz_read_hull", "rz_enable_aabb", "rz_read_aabb",
    "rz_instance_range", "rz_map_pose", "rz_commit_pose", "rz_time_span",
]
# symbols a library older than the current ABI lacks (ABI 5: rz_gather_chunk; 6: rz_device_numa_node; 7: the last four)
OPTIONAL_SYMBOLS = {"rz_gather_chunk", "rz_device_numa_node", "rz_instance_range", "rz_map_pose", "rz_commit_pose", "rz_time_span"}
POSE_WORLD16, POSE_ROWS12 = 0, 1


class RzAnimation(ctypes.Structure):
    _fields_ = [("n_bone_tracks", ctypes.c_uint32), ("track_bone", ctypes.POINTER(ctypes.c_int32)),
                ("key_off", ctypes.POINTER(ctypes.c_uint32)), ("key_frame", ctypes.POINTER(ctypes.c_float)),
                ("key_rot4", ctypes.POINTER(ctypes.c_float)), ("key_pos3", ctypes.POINTER(ctypes.c_float)),
                ("key_interp16", ctypes.POINTER(ctypes.c_uint8)), ("n_morph_tracks", ctypes.c_uint32),
                ("mkey_off", ctypes.POINTER(ctypes.c_uint32)), ("mkey_frame", ctypes.POINTER(ctypes.c_float)),
                ("mkey_weight", ctypes.POINTER(ctypes.c_float)), ("feed_off", ctypes.POINTER(ctypes.c_uint32)),
                ("feed_track", ctypes.POINTER(ctypes.c_int32)), ("feed_ratio", ctypes.POINTER(ctypes.c_float))]


class RzTiming(ctypes.Structure):
    _fields_ = [("frame_ms", ctypes.c_double), ("deform_kernel_ms", ctypes.c_double),
                ("prep_kernel_ms", ctypes.c_double), ("verts_per_frame", ctypes.c_uint64),
                ("algorithmic_bytes_per_frame", ctypes.c_uint64), ("frames", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32)]


class RzTuneEntry(ctypes.Structure):
    _fields_ = [("morph_split", ctypes.c_int), ("grid_cap", ctypes.c_int), ("inst_loop", ctypes.c_int),
                ("eff_split", ctypes.c_int), ("eff_grid", ctypes.c_int), ("eff_inst_group", ctypes.c_int),
                ("same_as", ctypes.c_int), ("ms", ctypes.c_float), ("ms_min", ctypes.c_float), ("ms_max", ctypes.c_float)]


class RzError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("reze_deform error %d: %s" % (code, message))
        self.code = code


_lib = None
_libs = {}
# the tools-only build that carries EVERY kernel variant (make -C reze-engine_amd/csrc variants): the parity tests reach the
# variants the product does not ship through it; nothing in the product, bench.py or host/ loads it
VARIANTS_LIB_PATH = os.path.join(os.path.dirname(_HERE), "tools", "variants", "libreze_deform_variants.so")


def load(path=None):
    """dlopen the in-tree HIP library (or, for tests / tools, another build of it given by `path`); raises (never falls
    back) when it is missing. Each build is bound once; their exported names are the same, so they are loaded RTLD_LOCAL
    and linked -Bsymbolic-functions: a call inside one build never lands in another."""
    global _lib
    if path is None:
        if _lib is not None:
            return _lib
        path = LIB_PATH
    path = os.path.abspath(path)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError("%s not built — run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    vp = ctypes.c_void_p
    fp = ctypes.POINTER(ctypes.c_float)
    u32 = ctypes.c_uint32
    L.rz_last_error.restype = ctypes.c_char_p
    L.rz_last_error.argtypes = []
    L.rz_abi_version.argtypes = []
    L.rz_device_count.argtypes = [ctypes.POINTER(ctypes.c_int)]
    if hasattr(L, "rz_device_numa_node"):      # (absent from libraries older than ABI 6)
        L.rz_device_numa_node.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.rz_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.rz_destroy.argtypes = [vp]
    L.rz_shard_range.argtypes = [u32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    if hasattr(L, "rz_gather_chunk"):          # (absent from libraries older than ABI 5, which tools/ab_inproc.py loads for A/B runs)
        L.rz_gather_chunk.argtypes = [u32, ctypes.c_int, ctypes.POINTER(u32)]
    L.rz_upload_mesh.argtypes = [vp, u32, fp, ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint8)]
    L.rz_upload_mesh_soa.argtypes = [vp, u32, fp, fp, ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint8)]
    L.rz_upload_skeleton.argtypes = [vp, u32, fp]
    L.rz_upload_morphs_dense.argtypes = [vp, u32, fp]
    L.rz_upload_morphs_sparse.argtypes = [vp, u32, ctypes.POINTER(u32), ctypes.POINTER(u32), fp]
    L.rz_set_instances.argtypes = [vp, u32]
    L.rz_set_pose.argtypes = [vp, fp, fp]
    i32p = ctypes.POINTER(ctypes.c_int32)
    L.rz_upload_skeleton_topology.argtypes = [vp, u32, i32p, fp, i32p, fp, ctypes.POINTER(ctypes.c_uint8)]
    L.rz_set_pose_local.argtypes = [vp, fp, fp, fp]
    L.rz_upload_animation.argtypes = [vp, ctypes.POINTER(RzAnimation)]
    L.rz_set_pose_sampled.argtypes = [vp, fp]
    L.rz_override_world.argtypes = [vp, u32, ctypes.POINTER(u32), ctypes.POINTER(u32), fp]
    L.rz_upload_bone_morphs.argtypes = [vp, u32, ctypes.POINTER(u32), ctypes.POINTER(u32), fp, fp]
    L.rz_read_world.argtypes = [vp, u32, fp]
    L.rz_deform.argtypes = [vp]
    L.rz_deform_n.argtypes = [vp, u32]
    L.rz_fork.argtypes = [vp, ctypes.POINTER(vp)]
    L.rz_deform_pair.argtypes = [vp, vp, u32]
    L.rz_sync.argtypes = [vp]
    L.rz_read.argtypes = [vp, u32, u32, u32, fp, fp]
    L.rz_read_palette.argtypes = [vp, u32, fp]
    L.rz_time_frames.argtypes = [vp, u32, ctypes.POINTER(RzTiming)]
    L.rz_autotune.argtypes = [vp, u32]
    L.rz_autotune_measure.argtypes = [vp, u32, ctypes.POINTER(RzTuneEntry), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.rz_autotune_pick.argtypes = [ctypes.POINTER(RzTuneEntry), ctypes.c_int]
    L.rz_autotune_apply.argtypes = [vp, ctypes.POINTER(RzTuneEntry)]
    L.rz_comm_info.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.rz_set_tuning.argtypes = [vp, ctypes.c_char_p, ctypes.c_int]
    L.rz_get_tuning.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    L.rz_output_ptrs.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(u32)]
    L.rz_comm_unique_id.argtypes = [ctypes.c_char_p]
    L.rz_rccl_info.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.rz_comm_init.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, u32]
    L.rz_allgather.argtypes = [vp, ctypes.c_int]
    L.rz_read_gathered.argtypes = [vp, u32, u32, fp, fp]
    L.rz_upload_edge_scale.argtypes = [vp, u32, fp]
    L.rz_read_hull.argtypes = [vp, u32, u32, u32, fp]
    L.rz_enable_aabb.argtypes = [vp, ctypes.c_int]
    L.rz_read_aabb.argtypes = [vp, u32, fp]
    L.rz_comm_init_all.argtypes = [ctypes.POINTER(vp), ctypes.c_int, u32]
    L.rz_allgather_all.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int]
    L.rz_gather_direct.argtypes = [ctypes.POINTER(vp), ctypes.c_int, u32, ctypes.c_int]
    L.rz_gather_fence.argtypes = [vp]
    if hasattr(L, "rz_map_pose"):              # (ABI 7)
        L.rz_instance_range.argtypes = [u32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(u32), ctypes.POINTER(u32)]
        L.rz_map_pose.argtypes = [vp, ctypes.c_int, ctypes.POINTER(fp), ctypes.POINTER(fp)]
        L.rz_commit_pose.argtypes = [vp]
        L.rz_time_span.argtypes = [vp, vp, u32, u32, ctypes.POINTER(ctypes.c_double)]
    for name in SYMBOLS:
        # (libraries older than the current ABI — tools/ab_inproc.py loads them side by side — lack the newer symbols: OPTIONAL_SYMBOLS)
        if name != "rz_last_error" and (name not in OPTIONAL_SYMBOLS or hasattr(L, name)):
            getattr(L, name).restype = ctypes.c_int
    _libs[path] = L
    if path == os.path.abspath(LIB_PATH):
        _lib = L
    return L


def _chk(code, L=None):
    if code != 0:
        raise RzError(code, (L or load()).rz_last_error().decode("utf-8", "replace"))


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def device_count():
    n = ctypes.c_int(0)
    rc = load().rz_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


def shard_range(v_total, nranks, rank):
    """Pure host helper (works without a GPU): (begin, count) of `rank`'s vertex shard."""
    b = ctypes.c_uint32(0)
    n = ctypes.c_uint32(0)
    _chk(load().rz_shard_range(int(v_total), int(nranks), int(rank), ctypes.byref(b), ctypes.byref(n)))
    return b.value, n.value


def instance_range(instances, nranks, rank):
    """Pure host helper: (begin, count) of the instances `rank` poses when a crowd is sharded along the instance axis."""
    b = ctypes.c_uint32(0)
    n = ctypes.c_uint32(0)
    _chk(load().rz_instance_range(int(instances), int(nranks), int(rank), ctypes.byref(b), ctypes.byref(n)))
    return b.value, n.value


def gather_chunk(v_total, nranks):
    """Pure host helper: the stride, in vertices, between two ranks' blocks of the gathered buffers."""
    c = ctypes.c_uint32(0)
    _chk(load().rz_gather_chunk(int(v_total), int(nranks), ctypes.byref(c)))
    return c.value


def comm_unique_id():
    buf = ctypes.create_string_buffer(128)
    _chk(load().rz_comm_unique_id(buf))
    return buf.raw


def rccl_info():
    """{path, version, reused}: the RCCL the library bound (reused = a copy the process had already loaded, e.g. PyTorch's)."""
    buf = ctypes.create_string_buffer(512)
    ver, reused = ctypes.c_int(0), ctypes.c_int(0)
    _chk(load().rz_rccl_info(buf, 512, ctypes.byref(ver), ctypes.byref(reused)))
    return {"path": buf.value.decode("utf-8", "replace"), "version": ver.value, "reused": bool(reused.value)}


def comm_init_all(contexts, v_total):
    """Single-process multi-GPU: contexts[r] is rank r (ncclCommInitAll)."""
    arr = (ctypes.c_void_p * len(contexts))(*[c._h for c in contexts])
    _chk(load().rz_comm_init_all(arr, len(contexts), int(v_total)))
    for c in contexts:
        c.v_total = int(v_total)


def device_numa_node(device=0, lib=None):
    """NUMA node of the host the device hangs off (rz_device_numa_node), -1 when the system does not say."""
    L = lib if lib is not None else load()
    n = ctypes.c_int(-1)
    _chk(L.rz_device_numa_node(int(device), ctypes.byref(n)), L)
    return n.value


def bind_to_device_node(device=0, lib=None):
    """Restrict the calling process to the cores of the device's NUMA node (what numactl --cpunodebind does): per-frame inputs cross the
    host link, and feeding a GPU from the other socket costs host time per HIP call and link bandwidth per pulled byte
    (include/reze_deform.h: rz_device_numa_node). Call it before contexts are created — pinned rings are first touched by the thread that
    creates them. Returns {"gpu_node": n, "cpus": "64-127,192-255"} or None when there is nothing to do (unknown node, one node,
    the node's cores are outside the affinity mask the process was given)."""
    node = device_numa_node(device, lib)
    if node < 0 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            text = f.read().strip()
        if len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()]) < 2:
            return None
    except OSError:
        return None
    cpus = set()
    for part in text.split(","):
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    cpus &= os.sched_getaffinity(0)
    if not cpus:
        return None
    os.sched_setaffinity(0, cpus)
    return {"gpu_node": node, "cpus": text}


def allgather_all(contexts, with_normals=False):
    arr = (ctypes.c_void_p * len(contexts))(*[c._h for c in contexts])
    _chk(load().rz_allgather_all(arr, len(contexts), 1 if with_normals else 0))


def gather_direct(contexts, v_total, root=0):
    """Peer-direct gather: every context's kernels store their shard straight into contexts[root]'s gathered
    buffer (no collective). contexts[r] holds shard r; read the whole mesh with contexts[root].read_gathered()."""
    arr = (ctypes.c_void_p * len(contexts))(*[c._h for c in contexts])
    _chk(load().rz_gather_direct(arr, len(contexts), int(v_total), int(root)))
    for c in contexts:
        c.v_total = int(v_total)


class DeformContext:
    """One GPU's deformation context (rz_ctx)."""

    def __init__(self, device=0, lib=None):
        """lib: a library returned by load(path) (tests: the all-variants build); default = the product library."""
        self._L = lib if lib is not None else load()
        h = ctypes.c_void_p()
        _chk(self._L.rz_create(int(device), ctypes.byref(h)), self._L)
        self._h = h
        self.V = 0
        self.B = 0
        self.M = 0
        self.I = 1

    def _chk(self, code):
        _chk(code, self._L)

    def close(self):
        if getattr(self, "_h", None):
            for f in list(getattr(self, "_forks", [])):       # forks borrow this context's static buffers: they go first
                f.close()
            self._L.rz_destroy(self._h)
            self._h = None
            lender = getattr(self, "_lender", None)
            if lender is not None and self in lender._forks:
                lender._forks.remove(self)

    def fork(self):
        """A second context on the same GPU that borrows this one's static data (no copy) and owns its own streams, pose
        slots and outputs: alternate frames between the two to keep two frames in flight (rz_fork)."""
        h = ctypes.c_void_p()
        self._chk(self._L.rz_fork(self._h, ctypes.byref(h)))
        f = DeformContext.__new__(DeformContext)
        f._L, f._h, f.V, f.B, f.M, f.I = self._L, h, self.V, self.B, self.M, self.I
        f._lender = self
        if not hasattr(self, "_forks"):
            self._forks = []
        self._forks.append(f)
        return f

    def deform_pair(self, other, frames):
        """`frames` frames alternating between this context and `other` (each on its own stream, into its own outputs)."""
        self._chk(self._L.rz_deform_pair(self._h, other._h, int(frames)))

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- static uploads ----
    def upload_mesh_interleaved(self, vertices8, joints4, weights4):
        v = _f32(vertices8).reshape(-1, 8)
        j = np.ascontiguousarray(joints4, dtype=np.uint16).reshape(-1, 4)
        w = np.ascontiguousarray(weights4, dtype=np.uint8).reshape(-1, 4)
        assert len(v) == len(j) == len(w)
        self._chk(self._L.rz_upload_mesh(self._h, len(v), _fptr(v), j.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)),
                                    w.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))))
        self.V = len(v)
        self.M = 0

    def upload_mesh(self, pos, nrm, joints4, weights4):
        p = _f32(pos).reshape(-1, 3)
        n = _f32(nrm).reshape(-1, 3)
        j = np.ascontiguousarray(joints4, dtype=np.uint16).reshape(-1, 4)
        w = np.ascontiguousarray(weights4, dtype=np.uint8).reshape(-1, 4)
        assert len(p) == len(n) == len(j) == len(w)
        self._chk(self._L.rz_upload_mesh_soa(self._h, len(p), _fptr(p), _fptr(n),
                                        j.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)),
                                        w.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))))
        self.V = len(p)
        self.M = 0

    def upload_skeleton(self, inverse_bind):
        ib = _f32(inverse_bind).reshape(-1, 16)
        self._chk(self._L.rz_upload_skeleton(self._h, len(ib), _fptr(ib)))
        self.B = len(ib)

    def upload_morphs_dense(self, deltas):
        if deltas is None or len(deltas) == 0:
            self._chk(self._L.rz_upload_morphs_dense(self._h, 0, None))
            self.M = 0
            return
        d = _f32(deltas)
        assert d.ndim == 3 and d.shape[1] == self.V and d.shape[2] == 3, d.shape
        self._chk(self._L.rz_upload_morphs_dense(self._h, d.shape[0], _fptr(d)))
        self.M = d.shape[0]

    def upload_morphs_sparse(self, morph_off, vert_idx, delta3):
        mo = np.ascontiguousarray(morph_off, dtype=np.uint32)
        vi = np.ascontiguousarray(vert_idx, dtype=np.uint32)
        d = _f32(delta3).reshape(-1, 3)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        self._chk(self._L.rz_upload_morphs_sparse(self._h, len(mo) - 1, mo.ctypes.data_as(u32p),
                                             vi.ctypes.data_as(u32p), _fptr(d)))
        self.M = len(mo) - 1

    def set_instances(self, n):
        self._chk(self._L.rz_set_instances(self._h, int(n)))
        self.I = int(n)

    # ---- per frame ----
    def set_pose(self, world, morph_weights=None):
        w = _f32(world).reshape(-1)
        assert w.size == self.I * self.B * 16, (w.size, self.I, self.B)
        if morph_weights is not None and self.M > 0:
            mw = _f32(morph_weights).reshape(-1)
            assert mw.size == self.I * self.M
            self._chk(self._L.rz_set_pose(self._h, _fptr(w), _fptr(mw)))
        else:
            self._chk(self._L.rz_set_pose(self._h, _fptr(w), None))

    def upload_skeleton_topology(self, parents, bind_translation, append_parent=None, append_ratio=None, append_move=None):
        par = np.ascontiguousarray(parents, dtype=np.int32)
        bind = _f32(bind_translation).reshape(-1, 3)
        i32p = ctypes.POINTER(ctypes.c_int32)
        ap = None if append_parent is None else np.ascontiguousarray(append_parent, dtype=np.int32)
        ar = None if append_ratio is None else _f32(append_ratio)
        am = None if append_move is None else np.ascontiguousarray(append_move, dtype=np.uint8)
        self._chk(self._L.rz_upload_skeleton_topology(
            self._h, len(par), par.ctypes.data_as(i32p), _fptr(bind),
            None if ap is None else ap.ctypes.data_as(i32p), None if ar is None else _fptr(ar),
            None if am is None else am.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))))

    def set_pose_local(self, local_rotations, morph_weights=None, local_translations=None):
        q = _f32(local_rotations).reshape(-1)
        assert q.size == self.I * self.B * 4, (q.size, self.I, self.B)
        t = None
        if local_translations is not None:
            t = _f32(local_translations).reshape(-1)
            assert t.size == self.I * self.B * 3
        mw = None
        if morph_weights is not None and self.M > 0:
            mw = _f32(morph_weights).reshape(-1)
            assert mw.size == self.I * self.M
        self._chk(self._L.rz_set_pose_local(self._h, _fptr(q), None if t is None else _fptr(t), None if mw is None else _fptr(mw)))

    def upload_animation(self, track_bone, key_off, key_frame, key_rot, key_pos, key_interp=None,
                         mkey_off=None, mkey_frame=None, mkey_weight=None, feed_off=None, feed_track=None, feed_ratio=None):
        """Flattened motion (rz_animation): bone tracks + optional morph tracks with their per-vertex-morph feeds."""
        keep = []

        def arr(x, dt, ct):
            if x is None:
                return None
            a = np.ascontiguousarray(x, dtype=dt).reshape(-1)
            keep.append(a)
            return a.ctypes.data_as(ctypes.POINTER(ct))
        a = RzAnimation()
        a.n_bone_tracks = len(track_bone)
        a.track_bone = arr(track_bone, np.int32, ctypes.c_int32)
        a.key_off = arr(key_off, np.uint32, ctypes.c_uint32)
        a.key_frame = arr(key_frame, np.float32, ctypes.c_float)
        a.key_rot4 = arr(key_rot, np.float32, ctypes.c_float)
        a.key_pos3 = arr(key_pos, np.float32, ctypes.c_float)
        a.key_interp16 = arr(key_interp, np.uint8, ctypes.c_uint8)
        a.n_morph_tracks = 0 if mkey_off is None else len(mkey_off) - 1
        a.mkey_off = arr(mkey_off, np.uint32, ctypes.c_uint32)
        a.mkey_frame = arr(mkey_frame, np.float32, ctypes.c_float)
        a.mkey_weight = arr(mkey_weight, np.float32, ctypes.c_float)
        a.feed_off = arr(feed_off, np.uint32, ctypes.c_uint32)
        a.feed_track = arr(feed_track, np.int32, ctypes.c_int32)
        a.feed_ratio = arr(feed_ratio, np.float32, ctypes.c_float)
        self._chk(self._L.rz_upload_animation(self._h, ctypes.byref(a)))

    def set_pose_sampled(self, frames):
        """One (fractional, 30 fps) frame per instance; bones, morph weights and the hierarchy are evaluated on the GPU."""
        f = _f32(np.atleast_1d(frames)).reshape(-1)
        assert f.size == self.I
        self._chk(self._L.rz_set_pose_sampled(self._h, _fptr(f)))

    def upload_bone_morphs(self, morph, bone, translation3, rotation4):
        """PMX bone morphs (type 2) for device-solved poses: entry k moves `bone[k]` by weight(morph[k]) * translation and
        right-multiplies its local rotation by slerp(identity, rotation, weight). Empty arrays clear."""
        m = np.ascontiguousarray(morph, dtype=np.uint32).reshape(-1)
        if m.size == 0:
            self._chk(self._L.rz_upload_bone_morphs(self._h, 0, None, None, None, None))
            return
        b = np.ascontiguousarray(bone, dtype=np.uint32).reshape(-1)
        t = _f32(translation3).reshape(-1)
        q = _f32(rotation4).reshape(-1)
        assert b.size == m.size and t.size == m.size * 3 and q.size == m.size * 4
        u32p = ctypes.POINTER(ctypes.c_uint32)
        self._chk(self._L.rz_upload_bone_morphs(self._h, int(m.size), m.ctypes.data_as(u32p), b.ctypes.data_as(u32p), _fptr(t), _fptr(q)))

    def override_world(self, bones, world16, instances=None):
        """Physics hand-off for device-solved poses (engine.ts:2379-2381): world matrices that replace the solved ones of
        (instance, bone) after the hierarchy solve, until the next call; empty `bones` clears."""
        b = np.ascontiguousarray(bones, dtype=np.uint32).reshape(-1)
        if b.size == 0:
            self._chk(self._L.rz_override_world(self._h, 0, None, None, None))
            return
        w = _f32(world16).reshape(-1)
        assert w.size == b.size * 16
        u32p = ctypes.POINTER(ctypes.c_uint32)
        ip = None
        if instances is not None:
            i = np.ascontiguousarray(instances, dtype=np.uint32).reshape(-1)
            assert i.size == b.size
            ip = i.ctypes.data_as(u32p)
        self._chk(self._L.rz_override_world(self._h, int(b.size), ip, b.ctypes.data_as(u32p), _fptr(w)))

    def read_world(self, instance=0):
        out = np.empty((self.B, 16), dtype=np.float32)
        self._chk(self._L.rz_read_world(self._h, int(instance), _fptr(out)))
        return out

    def upload_edge_scale(self, edge):
        if edge is None:
            self._chk(self._L.rz_upload_edge_scale(self._h, 0, None))
            return
        e = _f32(edge).reshape(-1)
        self._chk(self._L.rz_upload_edge_scale(self._h, len(e), _fptr(e)))

    def read_hull(self, instance=0, v0=0, n=None):
        n = self.V - v0 if n is None else n
        out = np.empty((n, 3), dtype=np.float32)
        self._chk(self._L.rz_read_hull(self._h, int(instance), int(v0), int(n), _fptr(out)))
        return out

    def enable_aabb(self, on=True):
        self._chk(self._L.rz_enable_aabb(self._h, 1 if on else 0))

    def read_aabb(self, instance=0):
        out = np.empty(6, dtype=np.float32)
        self._chk(self._L.rz_read_aabb(self._h, int(instance), _fptr(out)))
        return out

    def frame_call(self, kind, primary, morph_weights=None, translations=None):
        """A zero-argument callable that does ONE per-frame upload + rz_deform through the raw C ABI with every array
        converted and every pointer built up front — for timing per-frame loops without numpy conversions in them
        (bench.py, tools/live_loop.py). kind: 'world' (rz_set_pose), 'local' (rz_set_pose_local), 'sampled'
        (rz_set_pose_sampled; `primary` = a [T, I] table of frame numbers that is cycled through). Returns (call, check)."""
        L, h = self._L, self._h
        keep = []

        def ptr(a):
            if a is None:
                return None
            a = _f32(a).reshape(-1)
            keep.append(a)
            return _fptr(a)
        bad = []
        if kind == "world":
            wp, mp = ptr(primary), ptr(morph_weights if self.M > 0 else None)

            def call():
                if L.rz_set_pose(h, wp, mp) or L.rz_deform(h):
                    bad.append(1)
        elif kind == "local":
            qp, tp, mp = ptr(primary), ptr(translations), ptr(morph_weights if self.M > 0 else None)

            def call():
                if L.rz_set_pose_local(h, qp, tp, mp) or L.rz_deform(h):
                    bad.append(1)
        else:
            table = [ptr(row) for row in np.atleast_2d(primary)]
            tick = [0]

            def call():
                tick[0] += 1
                if L.rz_set_pose_sampled(h, table[tick[0] % len(table)]) or L.rz_deform(h):
                    bad.append(1)

        def check():
            if bad:
                raise RzError(-1, "a frame call failed: " + L.rz_last_error().decode("utf-8", "replace"))
        call.keep = keep
        return call, check

    def map_pose(self, layout=POSE_WORLD16):
        """rz_map_pose: (matrices, morph_weights) as numpy views OVER the pinned ring slot the next pose upload would have copied into —
        [I, B, 16] or (POSE_ROWS12) [I, B, 12] floats, and [I, M] floats or None. Write them in place, then commit_pose(). The views
        die with the commit (or the next pose call)."""
        mp, wp = ctypes.POINTER(ctypes.c_float)(), ctypes.POINTER(ctypes.c_float)()
        self._chk(self._L.rz_map_pose(self._h, int(layout), ctypes.byref(mp), ctypes.byref(wp)))
        per = 12 if layout == POSE_ROWS12 else 16
        mats = np.ctypeslib.as_array(mp, shape=(self.I, self.B, per))
        mw = np.ctypeslib.as_array(wp, shape=(self.I, self.M)) if (self.M > 0 and wp) else None
        return mats, mw

    def commit_pose(self):
        self._chk(self._L.rz_commit_pose(self._h))

    def mapped_frame_call(self, layout, fill=None):
        """Like frame_call, for caller-written poses: ONE rz_map_pose + fill(matrix pointer as int, bytes) + rz_commit_pose + rz_deform
        through the raw C ABI. `fill` stands in for the caller's pose solve writing its matrices in place (None: nothing is written — the
        protocol's own cost). Returns (call, check)."""
        L, h = self._L, self._h
        mp, wp = ctypes.POINTER(ctypes.c_float)(), ctypes.POINTER(ctypes.c_float)()
        mref, wref = ctypes.byref(mp), ctypes.byref(wp)
        nbytes = self.I * self.B * (48 if layout == POSE_ROWS12 else 64)
        bad = []

        def call():
            if L.rz_map_pose(h, layout, mref, wref):
                bad.append(1)
                return
            if fill is not None:
                fill(ctypes.cast(mp, ctypes.c_void_p).value, nbytes)
            if L.rz_commit_pose(h) or L.rz_deform(h):
                bad.append(1)

        def check():
            if bad:
                raise RzError(-1, "a mapped frame call failed: " + L.rz_last_error().decode("utf-8", "replace"))
        return call, check

    def time_span(self, frames, other=None, lead=0):
        """rz_time_span: ms between two events on the stream around `frames` back-to-back frames (with `other`, a fork: alternating),
        `lead` untimed frames in front of the opening event."""
        ms = ctypes.c_double(0.0)
        self._chk(self._L.rz_time_span(self._h, other._h if other is not None else None, int(lead), int(frames), ctypes.byref(ms)))
        return ms.value

    def deform(self):
        self._chk(self._L.rz_deform(self._h))

    def deform_n(self, frames):
        self._chk(self._L.rz_deform_n(self._h, int(frames)))

    def sync(self):
        self._chk(self._L.rz_sync(self._h))

    def read(self, instance=0, v0=0, n=None):
        n = self.V - v0 if n is None else n
        pos = np.empty((n, 3), dtype=np.float32)
        nrm = np.empty((n, 3), dtype=np.float32)
        self._chk(self._L.rz_read(self._h, int(instance), int(v0), int(n), _fptr(pos), _fptr(nrm)))
        return pos, nrm

    def read_palette(self, instance=0):
        out = np.empty((self.B, 12), dtype=np.float32)
        self._chk(self._L.rz_read_palette(self._h, int(instance), _fptr(out)))
        return out

    def autotune(self, frames=0):
        """Setup-time search over launch shapes with the current mesh / morphs / pose (rz_autotune)."""
        self._chk(self._L.rz_autotune(self._h, int(frames)))
        return {k: self.get_tuning(k) for k in ("effective_split", "effective_grid", "effective_inst_group")}

    _TUNE_FIELDS = ("morph_split", "grid_cap", "inst_loop", "eff_split", "eff_grid", "eff_inst_group", "same_as", "ms", "ms_min", "ms_max")

    def autotune_measure(self, frames=0):
        """rz_autotune_measure: the candidate table (list of dicts; entry 0 = the heuristic plan), nothing adopted."""
        tab = (RzTuneEntry * 32)()
        n = ctypes.c_int(0)
        self._chk(self._L.rz_autotune_measure(self._h, int(frames), tab, 32, ctypes.byref(n)))
        return [{k: getattr(tab[i], k) for k in self._TUNE_FIELDS} for i in range(n.value)]

    def autotune_pick(self, table):
        """rz_autotune_pick on a (possibly rank-reduced) table: entry 0 unless something beats it by >= 2 % with its slowest round under entry 0's fastest."""
        tab = (RzTuneEntry * len(table))()
        for i, e in enumerate(table):
            for k in self._TUNE_FIELDS:
                setattr(tab[i], k, e[k])
        return int(self._L.rz_autotune_pick(tab, len(table)))

    def autotune_apply(self, entry):
        e = RzTuneEntry()
        for k in self._TUNE_FIELDS:
            setattr(e, k, entry[k])
        self._chk(self._L.rz_autotune_apply(self._h, ctypes.byref(e)))

    def comm_info(self):
        """{count, user_rank} as the RCCL communicator reports them (ncclCommCount / ncclCommUserRank)."""
        n, u = ctypes.c_int(0), ctypes.c_int(-1)
        self._chk(self._L.rz_comm_info(self._h, ctypes.byref(n), ctypes.byref(u)))
        return {"comm_count": n.value, "comm_user_rank": u.value}

    def kernel_name(self):
        """The dominant kernel the CURRENT plan launches, spelled like rocprofv3's kernel trace spells it."""
        g = self.get_tuning
        tf = lambda k: "true" if g(k) else "false"  # noqa: E731
        if g("effective_poses_per_wg") > 0:
            return "rz_skin_instances_reg_kernel<8, %s>" % tf("effective_nt_store")
        if g("effective_inst_group") > 0:
            try:
                fused = g("effective_closure_bones") > 0        # the hierarchy solved in the crowd kernel's front (ABI 5)
            except RzError:
                fused = False
            if fused:
                return "rz_skin_instances_fk_kernel<%d, %s>" % (g("effective_inst_block"), tf("effective_nt_store"))
            return "rz_skin_instances_kernel<%d, %s, %s>" % (g("effective_inst_block"), tf("effective_nt_store"), tf("effective_subsets"))
        mode = g("morph_mode")
        s_ = g("effective_split")
        try:
            var = ", %d" % g("effective_variant")       # (ABI 6: the kernel variant without the fused consumers / with the specialised solve)
        except RzError:
            var = ""
        if mode == 1:
            return "rz_deform_dense_kernel<%d, %d, %s, %s, %s, %s%s>" % (s_, 8 if g("effective_unroll") >= 8 else 4, tf("effective_nt"), tf("effective_nt_store"),
                                                                         tf("effective_geo"), tf("effective_fast"), var)
        return "rz_deform_small_kernel<%d, %d, %s, %s, %s%s>" % (4 if s_ >= 4 else 1, mode, tf("effective_nt_store"), tf("effective_geo"), tf("effective_fast"), var)

    def time_frames(self, frames):
        t = RzTiming()
        self._chk(self._L.rz_time_frames(self._h, int(frames), ctypes.byref(t)))
        return {k: getattr(t, k) for k, _ in RzTiming._fields_ if k != "reserved"}

    def set_tuning(self, **kw):
        for k, v in kw.items():
            self._chk(self._L.rz_set_tuning(self._h, k.encode(), int(v)))

    def get_tuning(self, key):
        v = ctypes.c_int(0)
        self._chk(self._L.rz_get_tuning(self._h, key.encode(), ctypes.byref(v)))
        return v.value

    def output_ptrs(self):
        p = ctypes.c_void_p()
        n = ctypes.c_void_p()
        vp = ctypes.c_uint32(0)
        self._chk(self._L.rz_output_ptrs(self._h, ctypes.byref(p), ctypes.byref(n), ctypes.byref(vp)))
        return p.value, n.value, vp.value

    # ---- multi-GPU ----
    def comm_init(self, nranks, rank, unique_id, v_total):
        assert len(unique_id) == 128
        self._chk(self._L.rz_comm_init(self._h, int(nranks), int(rank), unique_id, int(v_total)))
        self.v_total = int(v_total)

    def allgather(self, with_normals=False):
        self._chk(self._L.rz_allgather(self._h, 1 if with_normals else 0))

    def gather_fence(self):
        """Make this (root) context's stream wait for the frames the other contributors have enqueued."""
        self._chk(self._L.rz_gather_fence(self._h))

    def read_gathered(self, v0=0, n=None):
        n = self.v_total - v0 if n is None else n
        pos = np.empty((n, 3), dtype=np.float32)
        nrm = np.empty((n, 3), dtype=np.float32)
        self._chk(self._L.rz_read_gathered(self._h, int(v0), int(n), _fptr(pos), _fptr(nrm)))
        return pos, nrm
