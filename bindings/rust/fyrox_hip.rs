//! Safe layer over `fyrox_hip_sys` (the generated `extern "C"` block): what sits behind the unchanged public
//! signatures of Fyrox on the skeletal-animation path.  Delivered as source for a Fyrox maintainer
//! (`fyrox-impl/src/scene/mesh/hip.rs`, cargo feature `hip-skinning`); the Rust toolchain is not part of this
//! repository's build image, so this file is not compiled or tested here -- the C ABI it calls is exercised
//! call for call by `tests/` through ctypes (`fyrox_amd/_native.py` declares the same prototypes).
//!
//! File:line references are to the Fyrox tree.
#![allow(dead_code)]

use super::hip_sys::*; // bindings/rust/fyrox_hip_sys.rs
use crate::core::algebra::{Matrix4, UnitQuaternion, Vector3};
use crate::scene::mesh::buffer::{VertexAttributeUsage, VertexBuffer};
use std::{ffi::c_void, ffi::CStr, ptr};

/// `fyx_status` as a Rust error.  `BoneIndex` is the slice-index panic of `scene/mesh/mod.rs:514`,
/// `MissingAttribute` is `VertexFetchError::NoSuchAttribute` (`scene/mesh/buffer.rs:1279`).
#[derive(Debug)]
pub enum HipError {
    InvalidArg(String),
    NoDevice,
    Hip(String),
    OutOfMemory,
    UnknownId,
    BoneIndex(String),
    MissingAttribute(String),
    Unsupported(String),
}

/// One per engine thread: the raw pointer makes it `!Send + !Sync`, as the C side requires
/// (`fyrox-impl/src/engine/mod.rs:1634-1733` is the only caller).
pub struct HipSkinning {
    ctx: *mut FyxCtx,
}

impl HipSkinning {
    pub fn new(device: i32) -> Result<Self, HipError> {
        let mut ctx = ptr::null_mut();
        match unsafe { fyx_init(&mut ctx, device) } {
            FYX_OK => Ok(Self { ctx }),
            _ => Err(HipError::NoDevice),
        }
    }

    /// One engine process, several GPUs: a context per device, joined into one RCCL communicator (`fyx_comm_init_all`;
    /// contexts[i] is rank i).  Every later call on them comes from this same thread, as the engine's update does.
    pub fn new_on_devices(devices: &[i32]) -> Result<Vec<Self>, HipError> {
        let mut all = Vec::with_capacity(devices.len());
        for d in devices {
            all.push(Self::new(*d)?);
        }
        let raw: Vec<*mut FyxCtx> = all.iter().map(|h| h.ctx).collect();
        let rc = unsafe { fyx_comm_init_all(raw.as_ptr(), raw.len() as i32) };
        check_rc(raw[0], rc)?;
        Ok(all)
    }

    /// The exchange step of a mesh sharded by vertex range over the GPUs of `new_on_devices` (`fyx_shard_vertex_range`):
    /// every GPU has skinned its shard in place into its own full-size streams; afterwards every GPU holds every shard.
    /// `pos` / `normal` / `tangent`: one device address per GPU, or an empty slice for a stream that is not exchanged.
    pub fn allgather_skinned_all(all: &mut [Self], n_verts: u32, pos: &[*mut f32], normal: &[*mut f32], tangent: &[*mut f32]) -> Result<(), HipError> {
        let raw: Vec<*mut FyxCtx> = all.iter().map(|h| h.ctx).collect();
        let p = |s: &[*mut f32]| if s.is_empty() { ptr::null() } else { s.as_ptr() };
        let rc = unsafe { fyx_allgather_skinned_all(raw.as_ptr(), raw.len() as i32, n_verts, p(pos), p(normal), p(tangent)) };
        check_rc(raw[0], rc)
    }

    /// The raw context for the other files of the shim (`fyrox_hip_flatten.rs`).
    pub(super) fn raw(&self) -> *mut FyxCtx {
        self.ctx
    }

    fn check(&self, rc: i32) -> Result<(), HipError> {
        check_rc(self.ctx, rc)
    }
}

/// `fyx_status` -> `Result`, with the context's last message.
pub(super) fn check_rc(ctx: *mut FyxCtx, rc: i32) -> Result<(), HipError> {
    if rc == FYX_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(fyx_last_error(ctx)) }.to_string_lossy().into_owned();
    Err(match rc {
        FYX_ERR_INVALID_ARG => HipError::InvalidArg(msg),
        FYX_ERR_NO_DEVICE => HipError::NoDevice,
        FYX_ERR_OOM => HipError::OutOfMemory,
        FYX_ERR_UNKNOWN_ID => HipError::UnknownId,
        FYX_ERR_BONE_INDEX => HipError::BoneIndex(msg),
        FYX_ERR_MISSING_ATTRIBUTE => HipError::MissingAttribute(msg),
        FYX_ERR_UNSUPPORTED => HipError::Unsupported(msg),
        _ => HipError::Hip(msg),
    })
}

impl HipSkinning {
    /// Upload (or re-upload) a surface's vertex buffer; call when `VertexBuffer::modifications_count()`
    /// (`buffer.rs:909`) changed.  The library de-interleaves on the GPU and keeps the interleaved bytes too.
    pub fn upload_vertex_buffer(&mut self, key: u64, vb: &VertexBuffer) -> Result<(), HipError> {
        // VertexBuffer::layout() (buffer.rs:980-982): the dense list of VertexAttribute {usage, offset, ..} (buffer.rs:175-202)
        let off = |u: VertexAttributeUsage| vb.layout().iter().find(|a| a.usage == u).map(|a| a.offset as i32).unwrap_or(-1);
        self.check(unsafe {
            fyx_mesh_upload(
                self.ctx,
                key,
                vb.raw_data().as_ptr(),
                vb.vertex_count(),
                vb.vertex_size() as u32,
                off(VertexAttributeUsage::Position),
                off(VertexAttributeUsage::Normal),
                off(VertexAttributeUsage::Tangent),
                off(VertexAttributeUsage::BoneWeight),
                off(VertexAttributeUsage::BoneIndices),
            )
        })
    }

    /// `Mesh::accurate_world_bounding_box`, skinned branch (`scene/mesh/mod.rs:487-522`): min / max of the skinned
    /// positions, reduced on the GPU.  `Matrix4<f32>` is column-major `[f32; 16]`: the palette is passed as is.
    pub fn skinned_aabb(&mut self, key: u64, bone_matrices: &[Matrix4<f32>]) -> Result<[f32; 6], HipError> {
        let mut b = [0f32; 6];
        self.check(unsafe {
            fyx_skinned_aabb(self.ctx, key, bone_matrices.as_ptr() as *const f32, bone_matrices.len() as u32, b.as_mut_ptr())
        })?;
        Ok(b)
    }

    /// The same box for EVERY instance of an instanced surface, device to device (`d_palettes`: `n_instances` palettes
    /// of `n_bones` matrices, e.g. what `HipAnimator::set_palette_output` registered; `d_out_boxes`: `n_instances` x
    /// `{min xyz, max xyz}`): what a culling pass over a crowd needs of `Mesh::accurate_world_bounding_box`
    /// (`scene/mesh/mod.rs:470-526`) without one call and one host round trip per instance.  Asynchronous.
    pub fn skinned_aabbs_device(
        &mut self,
        key: u64,
        d_palettes: *const f32,
        n_bones: u32,
        n_instances: u32,
        d_out_boxes: *mut f32,
    ) -> Result<(), HipError> {
        self.check(unsafe { fyx_skinned_aabb_device(self.ctx, key, d_palettes, n_bones, n_instances, d_out_boxes) })
    }

    /// Pipelined frames: whole frames alternate between two streams of the library, so frame n+1's pose kernels run beside
    /// frame n's skinning.  The caller then alternates two palette buffers per animator: a pose update must not be given a
    /// palette buffer that a skinning launch issued since the previous pose update reads ("Pipelined frames").
    pub fn set_pipelined(&mut self, on: bool) -> Result<(), HipError> {
        self.check(unsafe { fyx_set_option(self.ctx, b"anim.overlap\0".as_ptr() as *const _, on as i32) })
    }

    /// Additive API next to `SurfaceData` (`scene/mesh/surface.rs:265`): skin every vertex into host vectors.
    pub fn skin_into(&mut self, key: u64, palette: &[Matrix4<f32>], out: &mut SkinnedVertices) -> Result<(), HipError> {
        let n = out.positions.len();
        debug_assert!(out.normals.len() == n && out.tangents.len() == n);
        self.check(unsafe {
            fyx_lbs_skin(
                self.ctx,
                key,
                palette.as_ptr() as *const f32,
                palette.len() as u32,
                1,
                out.positions.as_mut_ptr() as *mut f32,
                out.normals.as_mut_ptr() as *mut f32,
                out.tangents.as_mut_ptr() as *mut f32,
                out.aabb.as_mut_ptr(),
            )
        })
    }

    /// Device-resident form: blend shapes (weights as `Mesh::collect_render_data` computes them,
    /// `scene/mesh/mod.rs:794-798`) and skinning straight into a vertex buffer with the surface's own layout,
    /// ready for the renderer's geometry cache (`renderer/cache/geometry.rs:84-93`).  Asynchronous; `join()`
    /// before the draw.
    pub fn skin_into_vertex_buffer(
        &mut self,
        key: u64,
        d_palette: *const f32,
        n_bones: u32,
        n_instances: u32,
        d_blend_shape_weights: *const f32,
        n_blend_shapes: u32,
        d_out_vertices: *mut u8,
    ) -> Result<(), HipError> {
        let desc = FyxSkinDesc {
            d_palette,
            n_bones,
            n_instances,
            d_blend_shape_weights,
            n_blend_shapes,
            d_out_pos: ptr::null_mut(),
            d_out_normal: ptr::null_mut(),
            d_out_tangent: ptr::null_mut(),
            d_out_vertices,
            out_stride: 0, // the surface's own vertex layout
            out_off_pos: -1,
            out_off_normal: -1,
            out_off_tangent: -1,
        };
        self.check(unsafe { fyx_lbs_skin_ex(self.ctx, key, &desc) })
    }

    /// One frame of every animated node of a scene (`Graph::update` -> `update_node`, `scene/graph/mod.rs:1415-1502`):
    /// `animators` are the ids of the `HipAnimator`s registered for the scene's `AnimationPlayer` /
    /// `AnimationBlendingStateMachine` nodes.  Same results as updating them one by one; one kernel launch per stage.
    pub fn update_scene(&mut self, animators: &[u64], dt: f32) -> Result<(), HipError> {
        self.check(unsafe { fyx_scene_update(self.ctx, animators.as_ptr(), animators.len() as u32, dt) })
    }

    /// Every skinned surface of a frame in one launch per layout class (`Mesh::collect_render_data`,
    /// `scene/mesh/mod.rs:774-802`, collects one `FyxSkinDesc` per surface instead of launching).
    pub fn skin_scene(&mut self, keys: &[u64], descs: &[FyxSkinDesc]) -> Result<(), HipError> {
        debug_assert_eq!(keys.len(), descs.len());
        self.check(unsafe { fyx_lbs_skin_ex_batch(self.ctx, keys.as_ptr(), descs.as_ptr(), keys.len() as u32) })
    }

    /// GPU-side join of every in-flight skinning launch with the context stream.
    pub fn join(&mut self) -> Result<(), HipError> {
        self.check(unsafe { fyx_join(self.ctx) })
    }
}

/// Which keys of a sampled curve the glTF importer keeps (`resource/gltf/simplify.rs:39-66`, run on every imported curve by
/// `gltf/animation.rs:155-163` with the binding's epsilon / max_step): indices into `x` / `y`.  No GPU, no context.
pub fn curve_simplify(x: &[f32], y: &[f32], epsilon: f32, max_step: f32) -> Result<Vec<u32>, HipError> {
    assert_eq!(x.len(), y.len());
    let mut out = vec![0u32; x.len().max(1)];
    let mut n = 0u32;
    let rc = unsafe { fyx_curve_simplify(x.as_ptr(), y.as_ptr(), x.len() as u32, epsilon, max_step, out.as_mut_ptr(), &mut n) };
    if rc != FYX_OK {
        return Err(if rc == FYX_ERR_OOM { HipError::OutOfMemory } else { HipError::InvalidArg("fyx_curve_simplify".to_string()) });
    }
    Ok(out[..n as usize].iter().copied().collect())
}

/// `BlendSpace::triangulate` (`machine/node/blendspace.rs:416-447`) for a shim that edits blend-space points outside the engine:
/// triangles as point indices, in the reference fixture's order (newest point first, counter-clockwise).  No GPU, no context.
pub fn blend_space_triangulate(points_xy: &[[f32; 2]]) -> Result<Vec<[u32; 3]>, HipError> {
    let cap = 4 * points_xy.len() as u32 + 4;
    let mut out = vec![[0u32; 3]; cap as usize];
    let mut n = 0u32;
    let rc = unsafe { fyx_blend_space_triangulate(points_xy.as_ptr() as *const f32, points_xy.len() as u32, out.as_mut_ptr() as *mut u32, cap, &mut n) };
    if rc != FYX_OK {
        return Err(if rc == FYX_ERR_OOM { HipError::OutOfMemory } else { HipError::InvalidArg("fyx_blend_space_triangulate: a coordinate is not finite".to_string()) });
    }
    Ok(out[..n as usize].iter().copied().collect())
}

impl Drop for HipSkinning {
    fn drop(&mut self) {
        unsafe { fyx_shutdown(self.ctx) }
    }
}

/// Output of `skin_into`.
pub struct SkinnedVertices {
    pub positions: Vec<Vector3<f32>>,
    pub normals: Vec<Vector3<f32>>,
    pub tangents: Vec<[f32; 4]>,
    pub aabb: [f32; 6],
}

/// Device-side output streams of one skinned surface (what `fyx_lbs_skin_device`, `fyx_lbs_skin_batch` and
/// `fyx_animator_set_skin_output` write): position / normal / tangent of `n_verts * n_instances` vertices, EACH STREAM ITS OWN
/// ALLOCATION (`fyx_malloc_streams`: ranges carved out of one block measured 77 - 80 us per crowd launch against 61 - 63,
/// profiles/r05_placement_pool/).  A renderer that draws frame n while frame n + 1 is skinned keeps two of these per surface.
pub struct DeviceSkinnedVertices {
    ctx: *mut FyxCtx,
    pub positions: *mut f32,
    pub normals: *mut f32,
    pub tangents: *mut f32,
}

impl DeviceSkinnedVertices {
    pub fn new(hip: &mut HipSkinning, n_verts: u32, n_instances: u32) -> Result<Self, HipError> {
        let n = n_verts as usize * n_instances as usize;
        let bytes: [usize; 3] = [n * 12, n * 12, n * 16];
        let mut ptrs: [*mut c_void; 3] = [std::ptr::null_mut(); 3];
        check_rc(hip.ctx, unsafe { fyx_malloc_streams(hip.ctx, 3, bytes.as_ptr(), ptrs.as_mut_ptr()) })?;
        Ok(Self { ctx: hip.ctx, positions: ptrs[0] as *mut f32, normals: ptrs[1] as *mut f32, tangents: ptrs[2] as *mut f32 })
    }
}

impl Drop for DeviceSkinnedVertices {
    fn drop(&mut self) {
        for p in [self.positions, self.normals, self.tangents] {
            unsafe { fyx_free(self.ctx, p as *mut c_void) };
        }
    }
}

/// N instances of one animated model: `AnimationPlayer` (`scene/animation/mod.rs:190-346`) and, optionally, the
/// `Machine` of an `AnimationBlendingStateMachine` (`scene/animation/absm.rs`), evaluated on the GPU.
/// Built once by flattening the engine's own objects: `HipAnimator::from_player` + `attach_machine` in
/// `fyrox_hip_flatten.rs`.
pub struct HipAnimator<'a> {
    hip: &'a mut HipSkinning,
    id: u64,
    n_instances: u32,
    /// signal index -> (Uuid, name), per animation: what `fyx_animation_pop_event` indices resolve to
    pub signal_names: Vec<Vec<(crate::core::uuid::Uuid, String)>>,
}

/// One `Property{..}` value as it comes back from `fyx_animator_read_properties`: the f32 lanes and the `TrackValue`
/// variant.  Re-wrapped into the engine's own `TrackValue`, so the numeric cast to the property's `ValueType`
/// (`TrackValue::apply_to_any`, `fyrox-animation/src/value.rs:234-352`) and the write through reflection
/// (`BoundValue::apply_to_object`, `value.rs:404-427`) are the engine's unchanged code.
pub fn track_value(v: &FyxPropertyValue) -> Option<crate::generic_animation::value::TrackValue> {
    use crate::core::algebra::{Quaternion, Vector2, Vector4};
    use crate::generic_animation::value::TrackValue;
    if v.present == 0 {
        return None;
    }
    let l = v.value;
    Some(match v.kind as i32 {
        FYX_VALUE_REAL => TrackValue::Real(l[0]),
        FYX_VALUE_VEC2 => TrackValue::Vector2(Vector2::new(l[0], l[1])),
        FYX_VALUE_VEC3 => TrackValue::Vector3(Vector3::new(l[0], l[1], l[2])),
        FYX_VALUE_VEC4 => TrackValue::Vector4(Vector4::new(l[0], l[1], l[2], l[3])),
        // lanes are (i, j, k, w); the library returns unit quaternions
        _ => TrackValue::UnitQuaternion(UnitQuaternion::new_unchecked(Quaternion::new(l[3], l[0], l[1], l[2]))),
    })
}

/// `Option<RootMotion>` as scripts read it (`fyrox-animation/src/lib.rs:325-336`).
pub struct HipRootMotion {
    pub delta_position: Vector3<f32>,
    pub delta_rotation: UnitQuaternion<f32>,
}

impl<'a> HipAnimator<'a> {
    /// Built by `HipAnimator::from_player` (`fyrox_hip_flatten.rs`), which flattens an `AnimationContainer`.
    pub(super) fn from_parts(hip: &'a mut HipSkinning, id: u64, n_instances: u32, signal_names: Vec<Vec<(crate::core::uuid::Uuid, String)>>) -> Self {
        Self { hip, id, n_instances, signal_names }
    }

    pub(super) fn raw(&self) -> *mut FyxCtx {
        self.hip.ctx
    }

    pub fn id(&self) -> u64 {
        self.id
    }

    /// `AnimationPlayer::update` with `auto_apply` (`scene/animation/mod.rs:340-346`).
    pub fn update_animations(&mut self, dt: f32) -> Result<(), HipError> {
        let rc = unsafe { fyx_animation_player_update(self.hip.ctx, self.id, dt) };
        self.hip.check(rc)
    }

    /// `AnimationBlendingStateMachine::update` (`scene/animation/absm.rs:311-326`).
    pub fn update_machine(&mut self, dt: f32) -> Result<(), HipError> {
        let rc = unsafe { fyx_absm_update(self.hip.ctx, self.id, dt) };
        self.hip.check(rc)
    }

    /// Every update of this animator also skins surface `mesh_key` with the palette output of `bones_id` into the three streams
    /// (`[n_instances][n_verts]`, null = not wanted; all null removes the entry): the character's frame is ONE call -- and, for one
    /// character, one launch (fyx_animator_set_skin_output, include/fyrox_hip.h).
    pub fn set_skin_output(&mut self, bones_id: u64, mesh_key: u64, d_pos: *mut f32, d_normal: *mut f32, d_tangent: *mut f32) -> Result<(), HipError> {
        let rc = unsafe { fyx_animator_set_skin_output(self.hip.ctx, self.id, bones_id, mesh_key, d_pos, d_normal, d_tangent) };
        self.hip.check(rc)
    }

    /// The update calls write this bone list's palettes (`[n_instances][n_bones]` column-major mat4) themselves from now on; null
    /// unregisters (fyx_animator_set_palette_output).
    pub fn set_palette_output(&mut self, bones_id: u64, d_out: *mut f32) -> Result<(), HipError> {
        let rc = unsafe { fyx_animator_set_palette_output(self.hip.ctx, self.id, bones_id, d_out) };
        self.hip.check(rc)
    }

    /// Two palette buffers for pipelined frames (`anim.overlap`): registered once, the frames of the library's two frame streams write
    /// one each, skin outputs read the frame's own (fyx_animator_set_palette_output_pair).
    pub fn set_palette_output_pair(&mut self, bones_id: u64, d_out: *mut f32, d_out_alt: *mut f32) -> Result<(), HipError> {
        let rc = unsafe { fyx_animator_set_palette_output_pair(self.hip.ctx, self.id, bones_id, d_out, d_out_alt) };
        self.hip.check(rc)
    }

    /// The buffer of a palette output that the most recent update call wrote (for a pair: the current frame's).
    pub fn current_palette(&mut self, bones_id: u64) -> Result<*mut f32, HipError> {
        let mut p: *mut f32 = std::ptr::null_mut();
        let rc = unsafe { fyx_animator_current_palette(self.hip.ctx, self.id, bones_id, &mut p) };
        self.hip.check(rc)?;
        Ok(p)
    }

    /// `Mesh::collect_render_data`'s `bone_matrices` (`scene/mesh/mod.rs:781-793`) for every instance, on the device.
    pub fn palette(&mut self, bones_id: u64, d_out: *mut f32) -> Result<(), HipError> {
        let rc = unsafe { fyx_animator_palette(self.hip.ctx, self.id, bones_id, d_out) };
        self.hip.check(rc)
    }

    /// `Animation::pop_event` (`fyrox-animation/src/lib.rs:680-682`).
    pub fn pop_event(&mut self, animation: u32, instance: u32) -> Result<Option<(crate::core::uuid::Uuid, String)>, HipError> {
        let mut signal = -1i32;
        let rc = unsafe { fyx_animation_pop_event(self.hip.ctx, self.id, animation, instance, &mut signal) };
        self.hip.check(rc)?;
        Ok(if signal < 0 { None } else { Some(self.signal_names[animation as usize][signal as usize].clone()) })
    }

    /// `machine.pose().root_motion()` of every instance after `update_machine`.
    pub fn machine_root_motion(&mut self) -> Result<Vec<Option<HipRootMotion>>, HipError> {
        let mut raw = vec![FyxRootMotion { delta_position: [0.0; 3], has: 0, delta_rotation: [0.0, 0.0, 0.0, 1.0] }; self.n_instances as usize];
        let rc = unsafe { fyx_absm_read_root_motion(self.hip.ctx, self.id, -1, raw.as_mut_ptr()) };
        self.hip.check(rc)?;
        Ok(raw
            .iter()
            .map(|r| {
                (r.has != 0).then(|| HipRootMotion {
                    delta_position: Vector3::new(r.delta_position[0], r.delta_position[1], r.delta_position[2]),
                    // nalgebra storage order (i, j, k, w); the library returns unit quaternions
                    delta_rotation: UnitQuaternion::new_unchecked(crate::core::algebra::Quaternion::new(
                        r.delta_rotation[3],
                        r.delta_rotation[0],
                        r.delta_rotation[1],
                        r.delta_rotation[2],
                    )),
                })
            })
            .collect())
    }
}
