/*
 * types.d.ts — the data shapes the host sources (host/src/*.ts) annotate with. Declarations only: nothing here reaches the shipped
 * JavaScript (tools/ts_erase.py drops `import type` lines with every other annotation). The shapes are the reference's
 * (engine/src/model.ts:6-68, vmd-loader.ts:4-24) plus what this build adds: morph sets, the flattened motion of the device sampler and
 * the raw N-API addon (csrc/napi_addon.c over include/reze_deform.h).
 */
import type { Quat, Vec3 } from './math'

export type NumArray = Float32Array | number[]
export type Quad = [number, number, number, number]
export type Triple = [number, number, number]

/** model.ts:24-34 */
export interface Bone {
  name: string
  parentIndex: number
  bindTranslation: Triple
  children: number[]
  appendParentIndex?: number
  appendRatio?: number
  appendRotate?: boolean
  appendMove?: boolean
}
/** model.ts:36-45 */
export interface Skeleton { bones: Bone[]; inverseBindMatrices: Float32Array }
export interface Skinning { joints: Uint16Array; weights: Uint8Array }
/** model.ts:52-59 */
export interface SkeletonRuntime {
  nameIndex: Record<string, number>
  localRotations: Float32Array
  localTranslations: Float32Array
  worldMatrices: Float32Array
  computedBones: boolean[]
}
/** model.ts:61-68 */
export interface RotTweenState {
  active: Uint8Array
  startQuat: Float32Array
  targetQuat: Float32Array
  startTimeMs: Float32Array
  durationMs: Float32Array
}
/** PMX bone morphs (type 2) and UV morphs (type 3), flattened by the loader; ascending morph index */
export interface BoneMorphEntries { morph: Uint32Array; bone: Uint32Array; translation: Float32Array; rotation: Float32Array }
export interface UvMorphEntries { morph: Uint32Array; vertex: Uint32Array; delta: Float32Array }
/** What PmxLoader.morphs() returns and Model keeps: vertex morphs as CSR over morphs (offsets / vertexIndex / deltas), group morphs as lists */
export interface MorphSet {
  names: string[]
  types: Uint8Array
  panels: Uint8Array
  groups: Array<Array<[number, number]> | null>
  offsets: Uint32Array
  vertexIndex: Uint32Array
  deltas: Float32Array
  boneEntries: BoneMorphEntries
  uvEntries: UvMorphEntries
}
export interface PosedLocals { rot: Float32Array; tra: Float32Array; moved: boolean }
export interface Material { name: string; edgeFlag: number; edgeSize: number; vertexCount: number; [key: string]: unknown }
export interface Texture { path: string; name: string }

/** vmd-loader.ts:4-24 (+ position and interpolation, which the reference drops, and the morph block it does not read) */
export interface BoneFrame { boneName: string; frame: number; rotation: Quat; position: Vec3; interpolation: Uint8Array }
export interface MorphFrame { morphName: string; frame: number; time: number; weight: number }
export interface VMDKeyFrame { time: number; boneFrames: BoneFrame[] }
export type VMDKeyFrames = VMDKeyFrame[] & { morphFrames?: MorphFrame[] }
export interface BoneSample { rotation: Quad; position: Triple }
/** VMDSampler.flatten(): the arrays of rz_animation (include/reze_deform.h) */
export interface FlatMotion {
  trackBone: Int32Array; keyOff: Uint32Array; keyFrame: Float32Array; keyRot: Float32Array; keyPos: Float32Array; keyInterp: Uint8Array
  mkeyOff?: Uint32Array; mkeyFrame?: Float32Array; mkeyWeight?: Float32Array
  feedOff?: Uint32Array; feedTrack?: Int32Array; feedRatio?: Float32Array
}

/** opaque rz_ctx handle of the addon */
export type DeformContext = unknown
export interface Timing { frameMs: number; deformKernelMs: number; prepKernelMs: number; vertsPerFrame: number; algorithmicBytesPerFrame: number; frames: number }
/** reze_deform.node — every function throws an Error carrying rz_last_error() on failure (csrc/napi_addon.c) */
export interface DeformAddon {
  abiVersion(): number
  deviceCount(): number
  deviceNumaNode(device: number): number
  create(device: number): DeformContext
  destroy(ctx: DeformContext): void
  fork(ctx: DeformContext): DeformContext
  shardRange(vTotal: number, nranks: number, rank: number): [number, number]
  instanceRange(instances: number, nranks: number, rank: number): [number, number]
  gatherChunk(vTotal: number, nranks: number): number
  uploadMesh(ctx: DeformContext, interleaved8: Float32Array, joints4: Uint16Array, weights4: Uint8Array): void
  uploadSkeleton(ctx: DeformContext, inverseBind: Float32Array): void
  uploadSkeletonTopology(ctx: DeformContext, parents: Int32Array, bind: Float32Array, appendParent: Int32Array | null, appendRatio: Float32Array | null, appendMove: Uint8Array | null): void
  uploadMorphsDense(ctx: DeformContext, morphs: number, deltas: Float32Array | null): void
  uploadMorphsSparse(ctx: DeformContext, offsets: Uint32Array, vertexIndex: Uint32Array, deltas: Float32Array): void
  uploadBoneMorphs(ctx: DeformContext, morph: Uint32Array | null, bone: Uint32Array | null, translation3: Float32Array | null, rotation4: Float32Array | null): void
  uploadAnimation(ctx: DeformContext, motion: FlatMotion): void
  uploadEdgeScale(ctx: DeformContext, edge: Float32Array | null): void
  enableAabb(ctx: DeformContext, on: boolean): void
  setInstances(ctx: DeformContext, count: number): void
  setPose(ctx: DeformContext, world: Float32Array, morphWeights: Float32Array | null): void
  setPoseLocal(ctx: DeformContext, localRotations: Float32Array, morphWeights: Float32Array | null, localTranslations?: Float32Array | null): void
  setPoseSampled(ctx: DeformContext, frames: Float32Array): void
  mapPose(ctx: DeformContext, layout?: 0 | 1): { matrices: Float32Array; morphWeights: Float32Array | null }
  commitPose(ctx: DeformContext): void
  overrideWorld(ctx: DeformContext, bones: Uint32Array, world16: Float32Array, instances: Uint32Array | null): void
  deform(ctx: DeformContext): void
  deformN(ctx: DeformContext, frames: number): void
  deformPair(ctx: DeformContext, fork: DeformContext, frames: number): void
  sync(ctx: DeformContext): void
  read(ctx: DeformContext, instance: number, v0: number, n: number, positions: Float32Array | null, normals: Float32Array | null): void
  readHull(ctx: DeformContext, instance: number, v0: number, n: number, hull: Float32Array): void
  readAabb(ctx: DeformContext, instance: number): Float32Array
  readWorld(ctx: DeformContext, instance: number, world16: Float32Array): void
  readPalette(ctx: DeformContext, instance: number, rows: Float32Array): void
  readGathered(ctx: DeformContext, v0: number, n: number, positions: Float32Array | null, normals: Float32Array | null): void
  timeFrames(ctx: DeformContext, frames: number): Timing
  timeSpan(ctx: DeformContext, fork: DeformContext | null, frames: number, lead?: number): number
  autotune(ctx: DeformContext, frames?: number): void
  setTuning(ctx: DeformContext, key: string, value: number): void
  getTuning(ctx: DeformContext, key: string): number
  commInitAll(ctxs: DeformContext[], vTotal: number): void
  allgatherAll(ctxs: DeformContext[], withNormals: boolean): void
  gatherDirect(ctxs: DeformContext[], vTotal: number, root: number): void
  gatherFence(root: DeformContext): void
}

/** the reference Physics' seam (engine.ts:2379-2381): may overwrite world matrices in place */
export interface PhysicsLike { step(dt: number, worldMatrices: Float32Array, inverseBindMatrices: Float32Array): void }
export interface EngineOptions {
  ambient?: number; bloomIntensity?: number; rimLightIntensity?: number; cameraDistance?: number; cameraTarget?: Vec3
  device?: number; devices?: number[]; deviceFK?: boolean; deviceSampling?: boolean; outline?: boolean; bounds?: boolean
  gather?: boolean | 'direct'; morphLayout?: 'sparse' | 'dense'; realtime?: boolean; physics?: PhysicsLike | null
  framesInFlight?: 1 | 2; autotune?: boolean
}
export interface EngineStats { fps: number; frameTime: number; gpuMemory: number; deformMs: number; vertsPerSec: number; hbmGBps: number }
export interface DeformedMesh { positions: Float32Array; normals: Float32Array }
export interface Bounds { min: number[]; max: number[] }
export interface Shard { ctx: DeformContext; begin: number; count: number; fork: DeformContext | null; last: DeformContext | null; flip: number }
export interface Timer { due: number; fn: () => void; id: number; handle: unknown }
