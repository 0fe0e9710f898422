"""Host-side vertex sharding for multi-GPU runs: one process per GPU, each owning a
contiguous vertex range of the mesh; static data is cut once at upload, per-frame inputs (world
matrices, morph weights) are replicated, and there is no exchange inside the frame. The layout of
the optional all-gather is fixed here too: rank r's vertices land at [r*chunk, r*chunk + count_r)."""
import numpy as np

from . import capi


def shard_of(v_total, world_size, rank):
    """(begin, count, chunk): this rank's range and the uniform chunk stride of the gathered buffer."""
    begin, count = capi.shard_range(v_total, world_size, rank)
    return begin, count, capi.gather_chunk(v_total, world_size)


def instances_of(instances, world_size, rank):
    """(begin, count): the instances this rank poses when a crowd (BASELINE config 4) is sharded along the instance axis — every rank
    holds the whole static mesh, there is no collective and no communicator (last sentence)."""
    return capi.instance_range(instances, world_size, rank)


def cut_mesh(mesh, deltas, begin, count):
    """Slice the per-vertex arrays of a synth.make_mesh() dict (+ dense deltas [M,V,3]) to one shard."""
    part = {k: np.ascontiguousarray(mesh[k][begin:begin + count]) for k in ("pos", "nrm", "joints", "weights")}
    part["inv_bind"] = mesh["inv_bind"]
    part["world"] = mesh["world"]
    d = None if deltas is None else np.ascontiguousarray(deltas[:, begin:begin + count])
    return part, d


def pad_to_chunk(arr, chunk):
    """Pad a [count, 3] result to the [chunk, 3] block a rank contributes to the all-gather."""
    out = np.zeros((chunk, arr.shape[1]), dtype=arr.dtype)
    out[:len(arr)] = arr
    return out


def gathered_to_mesh(gathered, v_total):
    """[world*chunk, 3] gathered buffer -> [v_total, 3]; shards are contiguous so it is a prefix."""
    return gathered[:v_total]
