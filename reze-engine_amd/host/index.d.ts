// Typings for the host package; shapes follow the reference's engine/src/index.ts:1-2 exports.
export class Vec3 {
  x: number; y: number; z: number
  constructor(x: number, y: number, z: number)
  add(o: Vec3): Vec3; subtract(o: Vec3): Vec3; scale(k: number): Vec3; dot(o: Vec3): number; cross(o: Vec3): Vec3
  length(): number; normalize(): Vec3; clone(): Vec3
}
export class Quat {
  x: number; y: number; z: number; w: number
  constructor(x: number, y: number, z: number, w: number)
  add(o: Quat): Quat; clone(): Quat; multiply(o: Quat): Quat; conjugate(): Quat; length(): number; normalize(): Quat
  rotateVec(v: Vec3): Vec3; rotate(v: Vec3): Vec3; toArray(): [number, number, number, number]; toEuler(): Vec3
  static slerp(a: Quat, b: Quat, t: number): Quat
  static fromEuler(rotX: number, rotY: number, rotZ: number): Quat
  static fromTo(from: Vec3, to: Vec3): Quat
}
export class Mat4 {
  values: Float32Array
  constructor(values: Float32Array)
  static identity(): Mat4
  static perspective(fov: number, aspect: number, near: number, far: number): Mat4
  static lookAt(eye: Vec3, target: Vec3, up: Vec3): Mat4
  static fromQuat(x: number, y: number, z: number, w: number): Mat4
  static fromPositionRotation(position: Vec3, rotation: Quat): Mat4
  static multiplyArrays(a: Float32Array, aOffset: number, b: Float32Array, bOffset: number, out: Float32Array, outOffset: number): void
  static toQuatFromArray(m: Float32Array, offset: number): Quat
  multiply(other: Mat4): Mat4; clone(): Mat4; getPosition(): Vec3; toQuat(): Quat; setIdentity(): this
  translateInPlace(tx: number, ty: number, tz: number): this; inverse(): Mat4
}
export interface EngineOptions {
  ambient?: number; bloomIntensity?: number; rimLightIntensity?: number; cameraDistance?: number; cameraTarget?: Vec3
  /** HIP device ordinal (default 0). */ device?: number
  /** Solve the bone hierarchy on the GPU (uploads local rotations instead of world matrices). */ deviceFK?: boolean
  /** Also write the outline pass's inverted hull every frame. */ outline?: boolean
  /** Also reduce the deformed mesh's bounding box every frame. */ bounds?: boolean
  /** One context per listed GPU; the mesh is vertex-sharded across them. */ devices?: number[]
  /** With deviceFK: seekFrame() samples the motion on the GPU (rz_upload_animation once, one float per frame). */ deviceSampling?: boolean
  /** Search launch shapes (morph split, workgroups per CU) once on the first rendered frame. */ autotune?: boolean
  /** 2: consecutive frames alternate between the context and a fork of it (own stream + outputs, shared static data), so frame f + 1 ramps up under frame f's tail. Single GPU, no gather. */ framesInFlight?: 1 | 2
  /** true: RCCL all-gather of the deformed mesh after every frame (distinct GPUs only); 'direct': every shard's kernel stores
   *  straight into the first GPU's gathered buffer over xGMI — no collective, GPUs may repeat in `devices`. */ gather?: boolean | 'direct'
  /** How PMX vertex morphs are laid out in HBM (default 'sparse'). */ morphLayout?: 'sparse' | 'dense'
  /** false: time only advances through step(timeMs). */ realtime?: boolean
  /** Physics hand-off (engine.ts:2379-2381), host-FK frames: step() may overwrite world matrices in place before they are uploaded. */
  physics?: { step(dt: number, boneWorldMatrices: Float32Array, boneInverseBindMatrices: Float32Array): void }
}
export interface EngineStats {
  fps: number; frameTime: number; gpuMemory: number
  deformMs: number; vertsPerSec: number; hbmGBps: number
}
export class Engine {
  constructor(canvas: unknown | null, options?: EngineOptions)
  init(): Promise<void>
  loadModel(path: string): Promise<void>
  loadAnimation(path: string): Promise<void>
  playAnimation(options?: { breathBones?: string[] | Record<string, number>; breathDuration?: number }): void
  stopAnimation(): void
  rotateBones(bones: string[], rotations: Quat[], durationMs?: number): void
  /** uv + UV morphs (PMX type 3), V x 2; a host-side sparse update — UVs never pass through the deformation kernel. */
  getMorphedUVs(): Float32Array
  /** Vertex morphs move vertices, bone morphs (PMX type 2) move bones before the hierarchy solve, group morphs feed both. */
  setMorphWeights(namesOrIndices: Array<string | number>, weights: number[]): void
  render(): void
  step(timeMs: number): void
  /** MMD-interpolated pose at a (fractional, 30 fps) frame; with a crowd (setInstanceCount) one frame per instance. */
  seekFrame(frame: number | ArrayLike<number>): void
  /** n independently posed copies of the model (needs { deviceFK, deviceSampling }, one GPU). */
  setInstanceCount(n: number): void
  /** Physics hand-off with { deviceFK }: world matrices (column-major 4x4 each) that replace the GPU-solved ones of the listed bones until the next call. */
  setBoneWorldOverrides(boneIndices: ArrayLike<number>, worldMatrices: ArrayLike<number>, instances?: ArrayLike<number>): void
  getDeformed(instance?: number): { positions: Float32Array; normals: Float32Array }
  getOutlineHull(): Float32Array
  getBounds(): { min: number[]; max: number[] }
  runRenderLoop(callback?: () => void): void
  stopRenderLoop(): void
  measure(frames?: number): { frameMs: number; deformKernelMs: number; prepKernelMs: number; vertsPerFrame: number; algorithmicBytesPerFrame: number; frames: number }
  getStats(): EngineStats
  dispose(): void
}
export class Model { [key: string]: any }
export class PmxLoader { static load(path: string): Promise<Model>; static loadFromBuffer(buf: ArrayBuffer | Uint8Array): Model }
export class VMDLoader { static load(path: string): Promise<any[]>; static loadFromBuffer(buf: ArrayBuffer | Uint8Array): any[] }
/** Frame-indexed MMD sampling of a loaded motion (rotation / position / morph keys); flatten() feeds the device sampler. */
export class VMDSampler {
  constructor(keyFrames: any[])
  lastFrame: number
  sampleBone(name: string, frame: number): { rotation: number[]; position: number[] } | null
  sampleMorph(name: string, frame: number): number | null
  boneNames(): string[]; morphNames(): string[]
  flatten(boneNameIndex: Record<string, number>, morphs: { names: string[]; types: ArrayLike<number>; groups: Array<Array<[number, number]> | null> } | null): Record<string, Int32Array | Uint32Array | Float32Array | Uint8Array>
}
