//! Flatteners: the engine's own animation objects -> the builder calls of the C ABI (`include/fyrox_hip.h`, second half).
//! Part of the shim of `fyrox_hip.rs` (`fyrox-impl/src/scene/mesh/hip_flatten.rs`, same cargo feature).  Like that file it
//! has never met a compiler in this repository's build image (no rustc); `tools/lint_rust_shim.py` (run by the CPU test
//! suite) checks every method, field, variant and path used here on a reference type against the Fyrox sources instead.
//!
//! What is flattened, and from where (file:line in the Fyrox tree):
//!   * the nodes of one animated model -> `fyx_rig_create`              (scene/base.rs:552,678,710; scene/transform.rs:196-407)
//!   * `AnimationTracksData`             -> `fyx_tracks_data_upload`      (fyrox-animation/src/lib.rs:71-110, track.rs:104-205, container.rs:103-160)
//!   * `AnimationContainer`              -> `fyx_animator_add_animation` + per-animation state  (lib.rs:269-291, :855, :951-1110)
//!   * `Machine`                         -> `fyx_machine_*` / `fyx_layer_*` / `fyx_state_*`      (machine/mod.rs:192-310, layer.rs:86-560)
//! Handles become dense indices: pools may have holes, so every pool is walked with `pair_iter()` and the handle -> index
//! maps are kept (`animation_index`, per-layer node / state maps) for the calls that take handles later.
#![allow(dead_code)]

use super::hip::{check_rc, HipAnimator, HipError, HipSkinning}; // bindings/rust/fyrox_hip.rs
use super::hip_sys::*; // bindings/rust/fyrox_hip_sys.rs
use crate::core::{algebra::Matrix3, math::curve::CurveKeyKind, pool::Handle};
use crate::generic_animation::{
    container::TrackValueKind,
    machine::{Parameter, PoseWeight},
    value::ValueBinding,
    AnimationTracksData,
};
use crate::graph::SceneGraph; // the trait that declares try_get_node (fyrox-graph/src/lib.rs; `pub use fyrox_graph as graph`, fyrox-impl/src/lib.rs:58)
use crate::scene::{
    animation::{
        absm::{Event, LogicNode, Machine, PoseNode, State, StateAction, Transition},
        Animation, AnimationContainer,
    },
    graph::Graph,
    node::Node,
};
use fxhash::FxHashMap;

/// Scene node handle -> index inside the rig registered with `fyx_rig_create`.
pub struct RigMap {
    pub rig_id: u64,
    pub index_of: FxHashMap<Handle<Node>, i32>,
}

impl RigMap {
    /// -1 for a handle outside the rig: the library treats a negative node as "no binding" / "invalid bone".
    pub fn node(&self, h: Handle<Node>) -> i32 {
        self.index_of.get(&h).copied().unwrap_or(-1)
    }
}

fn v3(v: &crate::core::algebra::Vector3<f32>) -> [f32; 3] {
    [v.x, v.y, v.z]
}

fn quat(q: &crate::core::algebra::UnitQuaternion<f32>) -> [f32; 4] {
    // nalgebra storage order (i, j, k, w)
    [q.coords.x, q.coords.y, q.coords.z, q.coords.w]
}

impl HipSkinning {
    /// `nodes`: the model's nodes parent-first (e.g. a depth-first walk from the model root); a node whose parent is not
    /// in the list gets parent -1 and multiplies by the identity, as `update_global_transform_recursively` does for an
    /// invalid parent handle (scene/graph/mod.rs:1210-1216).
    pub fn create_rig(&mut self, rig_id: u64, graph: &Graph, nodes: &[Handle<Node>]) -> Result<RigMap, HipError> {
        let mut index_of = FxHashMap::default();
        for (i, h) in nodes.iter().enumerate() {
            index_of.insert(*h, i as i32);
        }
        let mut parent = Vec::with_capacity(nodes.len());
        let mut transforms = Vec::with_capacity(nodes.len());
        let mut inv_bind = Vec::with_capacity(nodes.len() * 16);
        for h in nodes {
            let node = graph.try_get_node(*h).map_err(|_| HipError::InvalidArg(format!("node {h} is not in the graph")))?;
            parent.push(index_of.get(&node.parent()).copied().unwrap_or(-1));
            let t = node.local_transform();
            // Transform keeps the INVERSE of the post-rotation's matrix (build_post_rotation_matrix, transform.rs:160-172);
            // the field is private, so it is rebuilt here with the same nalgebra calls
            let por: Matrix3<f32> = t
                .post_rotation()
                .to_rotation_matrix()
                .matrix()
                .try_inverse()
                .unwrap_or_else(Matrix3::identity);
            let mut post = [0f32; 9];
            post.copy_from_slice(por.as_slice()); // column-major, as calculate_local_transform indexes it
            transforms.push(FyxTransform {
                local_position: v3(t.position()),
                local_rotation: quat(t.rotation()),
                local_scale: v3(t.scale()),
                pre_rotation: quat(t.pre_rotation()),
                post_rotation_matrix: post,
                rotation_offset: v3(t.rotation_offset()),
                rotation_pivot: v3(t.rotation_pivot()),
                scaling_offset: v3(t.scaling_offset()),
                scaling_pivot: v3(t.scaling_pivot()),
            });
            inv_bind.extend_from_slice(node.inv_bind_pose_transform().as_slice()); // Matrix4<f32>: column-major [f32; 16]
        }
        let rc = unsafe { fyx_rig_create(self.raw(), rig_id, nodes.len() as u32, parent.as_ptr(), transforms.as_ptr(), inv_bind.as_ptr()) };
        check_rc(self.raw(), rc)?;
        Ok(RigMap { rig_id, index_of })
    }

    /// `Surface::bones` (scene/mesh/surface.rs:1255) as a bone list of the rig; an invalid handle becomes -1 (identity
    /// matrix, scene/mesh/mod.rs:789-791).
    pub fn create_bone_list(&mut self, bones_id: u64, rig: &RigMap, bones: &[Handle<Node>]) -> Result<(), HipError> {
        let idx: Vec<i32> = bones.iter().map(|b| rig.node(*b)).collect();
        let rc = unsafe { fyx_bone_list_create(self.raw(), bones_id, rig.rig_id, idx.len() as u32, idx.as_ptr()) };
        check_rc(self.raw(), rc)
    }

    /// One `AnimationTracksData` resource (shared by any number of animations).  Keys are sent in the order the `Curve`s
    /// hold them: sorted by location (`Curve::from` / `add_key`, fyrox-math/src/curve.rs:170-236) -- the library refuses
    /// anything else.
    pub fn upload_tracks(&mut self, tracks_id: u64, data: &AnimationTracksData) -> Result<(), HipError> {
        let mut descs = Vec::with_capacity(data.tracks.len());
        let (mut loc, mut val, mut kind, mut lt, mut rt) = (Vec::new(), Vec::new(), Vec::new(), Vec::new(), Vec::new());
        let mut property_ids: FxHashMap<String, i32> = FxHashMap::default();
        for track in data.tracks() {
            let binding = match track.value_binding() {
                ValueBinding::Position => FYX_BIND_POSITION,
                ValueBinding::Scale => FYX_BIND_SCALE,
                ValueBinding::Rotation => FYX_BIND_ROTATION,
                // a property is addressed by a small integer id; the shim owns the name <-> id table of the tracks data
                ValueBinding::Property { name, .. } => {
                    let next = property_ids.len() as i32;
                    FYX_BIND_PROPERTY0 + *property_ids.entry(name.to_string()).or_insert(next)
                }
            };
            let container = track.data_container();
            let k = match container.value_kind() {
                TrackValueKind::Real => FYX_KIND_REAL,
                TrackValueKind::Vector2 => FYX_KIND_VEC2,
                TrackValueKind::Vector3 => FYX_KIND_VEC3,
                TrackValueKind::Vector4 => FYX_KIND_VEC4,
                TrackValueKind::UnitQuaternionEuler => FYX_KIND_QUAT_EULER,
                TrackValueKind::UnitQuaternion => FYX_KIND_QUAT,
            };
            let curves = container.curves_ref();
            if curves.len() > 4 {
                return Err(HipError::Unsupported("a track with more than four curves".into()));
            }
            let mut n_keys = [0u32; 4];
            for (c, curve) in curves.iter().enumerate() {
                n_keys[c] = curve.keys().len() as u32;
                for key in curve.keys() {
                    loc.push(key.location);
                    val.push(key.value);
                    match key.kind {
                        CurveKeyKind::Constant => {
                            kind.push(FYX_KEY_CONSTANT as u8);
                            lt.push(0.0);
                            rt.push(0.0);
                        }
                        CurveKeyKind::Linear => {
                            kind.push(FYX_KEY_LINEAR as u8);
                            lt.push(0.0);
                            rt.push(0.0);
                        }
                        CurveKeyKind::Cubic { left_tangent, right_tangent } => {
                            kind.push(FYX_KEY_CUBIC as u8);
                            lt.push(left_tangent);
                            rt.push(right_tangent);
                        }
                    }
                }
            }
            descs.push(FyxTrackDesc { binding, kind: k, n_curves: curves.len() as u32, curve_n_keys: n_keys });
        }
        let rc = unsafe {
            fyx_tracks_data_upload(
                self.raw(),
                tracks_id,
                descs.len() as u32,
                descs.as_ptr(),
                loc.len() as u32,
                loc.as_ptr(),
                val.as_ptr(),
                kind.as_ptr(),
                lt.as_ptr(),
                rt.as_ptr(),
            )
        };
        check_rc(self.raw(), rc)
    }
}

/// The scalar state of one `Animation` as the shim last saw it (see `push_animations` / `pull_animations`).
#[derive(Clone, Copy, PartialEq)]
pub struct AnimationShadow {
    pub speed: f32,
    pub looped: bool,
    pub enabled: bool,
    pub slice: (f32, f32),
    pub time: f32,
}

impl AnimationShadow {
    pub fn of(animation: &Animation) -> Self {
        let slice = animation.time_slice();
        Self {
            speed: animation.speed(),
            looped: animation.is_loop(),
            enabled: animation.is_enabled(),
            slice: (slice.start, slice.end),
            time: animation.time_position(),
        }
    }
}

/// What `from_player` returns besides the animator: the handle -> index tables the per-frame calls need.
pub struct AnimatorMaps {
    /// `Handle<Animation>` -> animation index inside the library (pool slots may be empty: indices are dense)
    pub animation_index: FxHashMap<Handle<Animation>, u32>,
    /// machine parameter name -> parameter index (filled by `attach_machine`)
    pub parameter_index: FxHashMap<String, u32>,
    /// per machine layer: the handle -> index tables of its three pools (filled by `attach_machine`); what
    /// `rebuild_machine` translates the run-time state through, and what turns the indices of `fyx_layer_get_state` and of
    /// the layer events back into handles
    pub layers: Vec<LayerMaps>,
    /// `machine_signature` of the definition that was sent last
    pub signature: u64,
    /// layer events (instance 0) that were still queued in the library when the definition was re-sent: `pop_layer_event`
    /// serves them first, so an edit loses none (the reference's queue lives in the layer object and survives edits)
    pub pending_layer_events: Vec<std::collections::VecDeque<Event>>,
}

/// The dense indices the library knows one layer's pool entries by.
#[derive(Default)]
pub struct LayerMaps {
    pub node_index: FxHashMap<Handle<PoseNode>, i32>,
    pub state_index: FxHashMap<Handle<State>, u32>,
    /// only transitions between two existing states are sent (see `attach_machine`)
    pub transition_index: FxHashMap<Handle<Transition>, u32>,
    /// the `BlendAnimationsByIndex` nodes: the only pose nodes with state of their own (node/blend.rs:260-264)
    pub by_index_nodes: Vec<Handle<PoseNode>>,
}

impl<'a> HipAnimator<'a> {
    /// `AnimationPlayer`'s container (scene/animation/mod.rs:285) -> an animator of `n_instances` copies of the rig.
    /// `tracks_id_of`: resource key of an animation's tracks data; every distinct key is uploaded once.
    pub fn from_player(
        hip: &'a mut HipSkinning,
        animator_id: u64,
        rig: &RigMap,
        animations: &AnimationContainer,
        n_instances: u32,
        mut tracks_id_of: impl FnMut(&Animation) -> u64,
    ) -> Result<(Self, AnimatorMaps), HipError> {
        let rc = unsafe { fyx_animator_create(hip.raw(), animator_id, rig.rig_id, n_instances) };
        check_rc(hip.raw(), rc)?;
        let mut uploaded: FxHashMap<u64, ()> = FxHashMap::default();
        let mut animation_index = FxHashMap::default();
        let mut signal_names = Vec::new();
        for (handle, animation) in animations.pair_iter() {
            let tracks_id = tracks_id_of(animation);
            let state = animation.tracks_data().state();
            let Some(data) = state.data_ref() else {
                continue; // not loaded: the reference's update_pose returns early too (lib.rs:896-899)
            };
            if uploaded.insert(tracks_id, ()).is_none() {
                hip.upload_tracks(tracks_id, data)?;
            }
            // TrackBinding per track id (lib.rs:855): target node and enabled flag; a track without a binding is skipped
            // by update_pose (lib.rs:903-905) -> target -1
            let mut target = Vec::with_capacity(data.tracks.len());
            let mut enabled = Vec::with_capacity(data.tracks.len());
            for track in data.tracks() {
                match animation.track_bindings().get(&track.id()) {
                    Some(b) => {
                        target.push(rig.node(b.target()));
                        enabled.push(b.is_enabled() as u8);
                    }
                    None => {
                        target.push(-1);
                        enabled.push(0);
                    }
                }
            }
            let mut index = 0u32;
            let rc = unsafe { fyx_animator_add_animation(hip.raw(), animator_id, tracks_id, target.as_ptr(), enabled.as_ptr(), &mut index) };
            check_rc(hip.raw(), rc)?;
            animation_index.insert(handle, index);
            // per-animation state, every instance alike (lib.rs:432-460, :713-748)
            let slice = animation.time_slice();
            unsafe {
                check_rc(hip.raw(), fyx_animation_set_loop(hip.raw(), animator_id, index, FYX_ALL_INSTANCES, animation.is_loop() as i32))?;
                check_rc(hip.raw(), fyx_animation_set_time_slice(hip.raw(), animator_id, index, FYX_ALL_INSTANCES, slice.start, slice.end))?;
                check_rc(hip.raw(), fyx_animation_set_time_position(hip.raw(), animator_id, index, FYX_ALL_INSTANCES, animation.time_position()))?;
                check_rc(hip.raw(), fyx_animation_set_speed(hip.raw(), animator_id, index, FYX_ALL_INSTANCES, animation.speed()))?;
                check_rc(hip.raw(), fyx_animation_set_enabled(hip.raw(), animator_id, index, FYX_ALL_INSTANCES, animation.is_enabled() as i32))?;
                check_rc(hip.raw(), fyx_animation_set_max_event_capacity(
                    hip.raw(),
                    animator_id,
                    index,
                    FYX_ALL_INSTANCES,
                    animation.get_max_event_capacity() as u32,
                ))?;
            }
            let mut names = Vec::new();
            for signal in animation.signals() {
                let mut s = 0u32;
                let rc = unsafe { fyx_animation_add_signal(hip.raw(), animator_id, index, signal.time, signal.enabled as i32, &mut s) };
                check_rc(hip.raw(), rc)?;
                names.push((signal.id, signal.name.clone()));
            }
            signal_names.push(names);
            if let Some(rm) = animation.root_motion_settings_ref() {
                let rc = unsafe {
                    fyx_animation_set_root_motion_settings(
                        hip.raw(),
                        animator_id,
                        index,
                        rig.node(rm.node),
                        rm.ignore_x_movement as i32,
                        rm.ignore_y_movement as i32,
                        rm.ignore_z_movement as i32,
                        rm.ignore_rotations as i32,
                    )
                };
                check_rc(hip.raw(), rc)?;
            }
        }
        let animator = HipAnimator::from_parts(hip, animator_id, n_instances, signal_names);
        Ok((animator, AnimatorMaps { animation_index, parameter_index: FxHashMap::default(), layers: Vec::new(), signature: 0, pending_layer_events: Vec::new() }))
    }

    /// Before the update: what game code did to the `Animation` objects since the last frame -- `set_speed`, `set_loop`,
    /// `set_enabled`, `set_time_slice`, `set_time_position` / `rewind` (lib.rs:432-470, :713-748) -- reaches the library.
    /// The library's copy is the one that ticks, so the objects are compared with `shadow` (what `pull_animations`
    /// wrote into them after the previous update, or what `from_player` sent): a field that differs was changed by the
    /// game.  Animations added to or removed from the container since `from_player` are the caller's to announce
    /// (`fyx_animator_add_animation` / `fyx_animator_remove_animation`); instance 0 is the engine's one instance.
    pub fn push_animations(&mut self, animations: &AnimationContainer, maps: &AnimatorMaps, shadow: &mut FxHashMap<Handle<Animation>, AnimationShadow>) -> Result<(), HipError> {
        let (ctx, id) = (self.raw(), self.id());
        for (h, animation) in animations.pair_iter() {
            let Some(index) = maps.animation_index.get(&h).copied() else {
                continue;
            };
            let now = AnimationShadow::of(animation);
            let was = shadow.get(&h).copied().unwrap_or(now);
            unsafe {
                if now.speed != was.speed {
                    check_rc(ctx, fyx_animation_set_speed(ctx, id, index, FYX_ALL_INSTANCES, now.speed))?;
                }
                if now.looped != was.looped {
                    check_rc(ctx, fyx_animation_set_loop(ctx, id, index, FYX_ALL_INSTANCES, now.looped as i32))?;
                }
                if now.slice != was.slice {
                    check_rc(ctx, fyx_animation_set_time_slice(ctx, id, index, FYX_ALL_INSTANCES, now.slice.0, now.slice.1))?;
                }
                if now.time != was.time {
                    check_rc(ctx, fyx_animation_set_time_position(ctx, id, index, FYX_ALL_INSTANCES, now.time))?;
                }
                if now.enabled != was.enabled {
                    check_rc(ctx, fyx_animation_set_enabled(ctx, id, index, FYX_ALL_INSTANCES, now.enabled as i32))?;
                }
            }
            shadow.insert(h, now);
        }
        Ok(())
    }

    /// After the update: the clocks and enabled flags the library advanced (ticks, `StateAction`s) go back into the
    /// objects, so game code that reads `animation.time_position()` / `is_enabled()` / `has_ended()` sees what the
    /// reference would show it.
    pub fn pull_animations(&mut self, animations: &mut AnimationContainer, maps: &AnimatorMaps, shadow: &mut FxHashMap<Handle<Animation>, AnimationShadow>) -> Result<(), HipError> {
        let (ctx, id) = (self.raw(), self.id());
        for (h, index) in maps.animation_index.iter() {
            let Some(animation) = animations.try_get_mut(*h).ok() else {
                continue;
            };
            let (mut time, mut enabled, mut ended) = (0.0f32, 0i32, 0i32);
            check_rc(ctx, unsafe { fyx_animation_get_state(ctx, id, *index, 0, &mut time, &mut enabled, &mut ended) })?;
            animation.set_time_position(time);
            animation.set_enabled(enabled != 0);
            shadow.insert(*h, AnimationShadow::of(animation));
        }
        Ok(())
    }

    /// The `Machine` of an `AnimationBlendingStateMachine` (scene/animation/absm.rs:240) on top of `from_player`.
    pub fn attach_machine(&mut self, machine: &Machine, rig: &RigMap, maps: &mut AnimatorMaps) -> Result<(), HipError> {
        let (ctx, id) = (self.raw(), self.id());
        for layer in machine.layers() {
            let mut li = 0u32;
            check_rc(ctx, unsafe { fyx_machine_add_layer(ctx, id, layer.weight(), &mut li) })?;
            let excluded: Vec<i32> = layer.mask().inner().iter().map(|h| rig.node(*h)).filter(|i| *i >= 0).collect();
            if !excluded.is_empty() {
                check_rc(ctx, unsafe { fyx_layer_set_mask(ctx, id, li, excluded.as_ptr(), excluded.len() as u32) })?;
            }
            // pose nodes: the pool is walked in slot order; children are referred to by handle, so the handle -> index map
            // is complete before any node is sent
            let mut node_index: FxHashMap<Handle<PoseNode>, i32> = FxHashMap::default();
            let mut by_index_nodes = Vec::new();
            for (n, (h, node)) in layer.nodes().pair_iter().enumerate() {
                node_index.insert(h, n as i32);
                if let PoseNode::BlendAnimationsByIndex(_) = node {
                    by_index_nodes.push(h);
                }
            }
            for (_, node) in layer.nodes().pair_iter() {
                let mut out = 0u32;
                let rc = match node {
                    PoseNode::PlayAnimation(play) => {
                        // a handle that does not resolve: the node keeps an empty pose (play.rs:93-99) -- an animation index
                        // past the end says the same to the library
                        let a = maps.animation_index.get(&play.animation).copied().unwrap_or(u32::MAX);
                        unsafe { fyx_layer_add_play_animation(ctx, id, li, a, &mut out) }
                    }
                    PoseNode::BlendAnimations(blend) => {
                        let mut sources = Vec::with_capacity(blend.pose_sources.len());
                        let mut wp = Vec::with_capacity(blend.pose_sources.len());
                        let mut wc = Vec::with_capacity(blend.pose_sources.len());
                        for p in blend.pose_sources.iter() {
                            sources.push(node_index.get(&p.pose_source).copied().unwrap_or(-1));
                            match &p.weight {
                                PoseWeight::Constant(w) => {
                                    wp.push(-1);
                                    wc.push(*w);
                                }
                                PoseWeight::Parameter(name) => {
                                    // a name that does not resolve weighs 0.0 (blend.rs:147-153)
                                    wp.push(resolve_parameter(ctx, id, machine, maps, name)?);
                                    wc.push(0.0);
                                }
                            }
                        }
                        unsafe { fyx_layer_add_blend_animations(ctx, id, li, sources.len() as u32, sources.as_ptr(), wp.as_ptr(), wc.as_ptr(), &mut out) }
                    }
                    PoseNode::BlendAnimationsByIndex(by_index) => {
                        let sources: Vec<i32> = by_index.inputs.iter().map(|i| node_index.get(&i.pose_source).copied().unwrap_or(-1)).collect();
                        let times: Vec<f32> = by_index.inputs.iter().map(|i| i.blend_time).collect();
                        let p = resolve_parameter(ctx, id, machine, maps, &by_index.index_parameter)?;
                        unsafe { fyx_layer_add_blend_animations_by_index(ctx, id, li, p, sources.len() as u32, sources.as_ptr(), times.as_ptr(), &mut out) }
                    }
                    PoseNode::BlendSpace(space) => {
                        let mut pts = Vec::with_capacity(space.points().len() * 2);
                        let mut sources = Vec::with_capacity(space.points().len());
                        for p in space.points() {
                            pts.push(p.position.x);
                            pts.push(p.position.y);
                            sources.push(node_index.get(&p.pose_source).copied().unwrap_or(-1));
                        }
                        // the Delaunay triangulation the engine caches on every point edit (blendspace.rs:246-270, :416-447)
                        let mut tris = Vec::with_capacity(space.triangles().len() * 3);
                        for t in space.triangles() {
                            tris.extend_from_slice(&[t[0], t[1], t[2]]);
                        }
                        let p = resolve_parameter(ctx, id, machine, maps, space.sampling_parameter())?;
                        unsafe {
                            fyx_layer_add_blend_space(
                                ctx,
                                id,
                                li,
                                p,
                                sources.len() as u32,
                                pts.as_ptr(),
                                sources.as_ptr(),
                                (tris.len() / 3) as u32,
                                tris.as_ptr(),
                                &mut out,
                            )
                        }
                    }
                };
                check_rc(ctx, rc)?;
            }
            // states, in slot order; the first one added becomes active (layer.rs:229-235), the entry state is set afterwards
            let mut state_index = FxHashMap::default();
            for (n, (h, _)) in layer.states().pair_iter().enumerate() {
                state_index.insert(h, n as u32);
            }
            for (_, state) in layer.states().pair_iter() {
                let mut si = 0u32;
                let root = node_index.get(&state.root).copied().unwrap_or(-1);
                check_rc(ctx, unsafe { fyx_layer_add_state(ctx, id, li, root, &mut si) })?;
                for (on_enter, actions) in [(1, &state.on_enter_actions), (0, &state.on_leave_actions)] {
                    for action in actions.iter() {
                        let index_of = |h: &Handle<Animation>| maps.animation_index.get(h).copied().unwrap_or(u32::MAX);
                        let rc = match &action.0 {
                            StateAction::None => FYX_OK,
                            StateAction::RewindAnimation(h) => unsafe {
                                fyx_state_add_action(ctx, id, li, si, on_enter, FYX_ACTION_REWIND_ANIMATION, index_of(h))
                            },
                            StateAction::EnableAnimation(h) => unsafe {
                                fyx_state_add_action(ctx, id, li, si, on_enter, FYX_ACTION_ENABLE_ANIMATION, index_of(h))
                            },
                            StateAction::DisableAnimation(h) => unsafe {
                                fyx_state_add_action(ctx, id, li, si, on_enter, FYX_ACTION_DISABLE_ANIMATION, index_of(h))
                            },
                            StateAction::EnableRandomAnimation(handles) => {
                                let list: Vec<u32> = handles.iter().map(index_of).collect();
                                unsafe { fyx_state_add_random_action(ctx, id, li, si, on_enter, list.as_ptr(), list.len() as u32) }
                            }
                        };
                        check_rc(ctx, rc)?;
                    }
                }
            }
            if let Some(e) = state_index.get(&layer.entry_state()) {
                check_rc(ctx, unsafe { fyx_layer_set_entry_state(ctx, id, li, *e) })?;
            }
            let mut transition_index = FxHashMap::default();
            for (th, transition) in layer.transitions().pair_iter() {
                let (Some(s), Some(d)) = (state_index.get(&transition.source()), state_index.get(&transition.dest())) else {
                    continue; // a transition between states that do not exist can never fire (layer.rs:606-611)
                };
                let mut code = Vec::new();
                encode_logic(transition.condition(), &mut code, ctx, id, machine, maps)?;
                let mut ti = 0u32;
                check_rc(ctx, unsafe {
                    fyx_layer_add_transition(ctx, id, li, *s, *d, transition.transition_time(), code.as_ptr(), code.len() as u32, &mut ti)
                })?;
                transition_index.insert(th, ti);
            }
            // The run-time state the objects carry (a scene loaded from a save file is in the middle of whatever it was
            // doing; from here on the library's copy is the one that moves): active state / transition (layer.rs:432-441),
            // every transition's blend factor (transition.rs:307; `elapsed_time` has no getter: blend_factor *
            // transition_time reproduces it to an ulp), prev_index / blend_time of the by-index nodes (node/blend.rs:260-264).
            let s = state_index.get(&layer.active_state()).map(|i| *i as i32).unwrap_or(-1);
            let t = transition_index.get(&layer.active_transition()).map(|i| *i as i32).unwrap_or(-1);
            check_rc(ctx, unsafe { fyx_layer_set_state(ctx, id, li, FYX_ALL_INSTANCES, s, t) })?;
            for (th, transition) in layer.transitions().pair_iter() {
                if let Some(ti) = transition_index.get(&th) {
                    let factor = transition.blend_factor();
                    check_rc(ctx, unsafe {
                        fyx_layer_set_transition_state(ctx, id, li, FYX_ALL_INSTANCES, *ti, factor * transition.transition_time(), factor)
                    })?;
                }
            }
            for h in by_index_nodes.iter() {
                if let PoseNode::BlendAnimationsByIndex(by_index) = &layer.nodes()[*h] {
                    let prev = by_index.prev_index.get();
                    check_rc(ctx, unsafe {
                        fyx_layer_set_node_state(
                            ctx,
                            id,
                            li,
                            FYX_ALL_INSTANCES,
                            node_index[h] as u32,
                            prev.is_some() as i32,
                            prev.unwrap_or(0),
                            by_index.blend_time.get(),
                        )
                    })?;
                }
            }
            maps.layers.push(LayerMaps { node_index, state_index, transition_index, by_index_nodes });
        }
        maps.signature = machine_signature(machine);
        Ok(())
    }

    /// Every frame, before `update_machine`: the definition again if the game has edited the `Machine` since it was sent
    /// (`machine_signature` differs -- nothing for the game to announce), then the parameter values.  With this the
    /// engine's own `AnimationBlendingStateMachine::update` body is the three calls `sync_machine`, `update_machine`
    /// and whatever reads the results.
    pub fn sync_machine(&mut self, machine: &Machine, rig: &RigMap, maps: &mut AnimatorMaps, n_instances: u32) -> Result<(), HipError> {
        if machine_signature(machine) != maps.signature {
            self.rebuild_machine(machine, rig, maps, n_instances)?;
        }
        self.sync_parameters(machine, maps)
    }

    /// After the game has edited the `Machine` in place -- `layers_mut`, `nodes_mut`, `transitions_mut`, `states_mut`,
    /// `Transition::set_condition`, `BlendSpace::set_points`, a pool entry freed ... (machine/mod.rs:280-312,
    /// layer.rs:412-525): the next `evaluate_pose` of the reference sees the edit with every other piece of run-time
    /// state untouched.  Here the definition is sent again and the run-time state -- which lives in the library, it is
    /// the library that evaluates -- is carried over BY HANDLE: active state / transition of every layer
    /// (layer.rs:103-109), `elapsed_time` / `blend_factor` of every transition (transition.rs:188-201), `prev_index` /
    /// `blend_time` of every `BlendAnimationsByIndex` node (node/blend.rs:260-264); the parameter values are the
    /// `ParameterContainer`'s own.  A handle that no longer resolves drops its state; an active state that is gone becomes
    /// `Handle::NONE`.  Layers are matched by position.  Layer events still queued in the library are moved into
    /// `maps.pending_layer_events` (as handles) and served by `pop_layer_event` first.  What edit happened is the game's knowledge: call this when it says so (the editor's
    /// commands, a script that rewires the graph) -- not every frame.
    pub fn rebuild_machine(&mut self, machine: &Machine, rig: &RigMap, maps: &mut AnimatorMaps, n_instances: u32) -> Result<(), HipError> {
        let (ctx, id) = (self.raw(), self.id());
        struct SavedLayer {
            known_states: Vec<Handle<State>>,
            active_state: Option<Handle<State>>,
            active_transition: Option<Handle<Transition>>,
            transitions: Vec<(Handle<Transition>, f32, f32)>,
            by_index: Vec<(Handle<PoseNode>, i32, u32, f32)>,
        }
        let mut saved: Vec<Vec<SavedLayer>> = Vec::new(); // [instance][layer]
        for instance in 0..n_instances {
            let mut per_layer = Vec::new();
            for (li, lm) in maps.layers.iter().enumerate() {
                let li = li as u32;
                let (mut s, mut t) = (-1i32, -1i32);
                check_rc(ctx, unsafe { fyx_layer_get_state(ctx, id, li, instance, &mut s, &mut t) })?;
                let mut transitions = Vec::new();
                for (h, index) in lm.transition_index.iter() {
                    let (mut elapsed, mut factor) = (0.0f32, 0.0f32);
                    check_rc(ctx, unsafe { fyx_layer_get_transition_state(ctx, id, li, instance, *index, &mut elapsed, &mut factor) })?;
                    transitions.push((*h, elapsed, factor));
                }
                let mut by_index = Vec::new();
                for h in lm.by_index_nodes.iter() {
                    let (mut has_prev, mut prev, mut time) = (0i32, 0u32, 0.0f32);
                    let node = lm.node_index[h] as u32;
                    check_rc(ctx, unsafe { fyx_layer_get_node_state(ctx, id, li, instance, node, &mut has_prev, &mut prev, &mut time) })?;
                    by_index.push((*h, has_prev, prev, time));
                }
                per_layer.push(SavedLayer {
                    known_states: lm.state_index.keys().copied().collect(),
                    active_state: lm.state_index.iter().find(|(_, i)| **i as i32 == s).map(|(h, _)| *h),
                    active_transition: lm.transition_index.iter().find(|(_, i)| **i as i32 == t).map(|(h, _)| *h),
                    transitions,
                    by_index,
                });
            }
            saved.push(per_layer);
        }
        // events still queued: out of the library (they do not survive the clear), into the shim's own queue, as handles.
        // KNOWN LOSS: only instance 0's events are kept -- the `Machine` this shim mirrors is ONE machine (instance 0 is the
        // engine's own `MachineLayer::pop_event` stream); the queued events of instances 1.. of a crowd animator are dropped by an
        // in-place edit.  The Python mirror (fyrox_amd/anim.py::rebuild_machine) keeps them per (layer, instance); a crowd front
        // end that needs them across edits must do the same (a queue per (layer, instance) and an instance argument to
        // pop_layer_event).  Not done here: this file has never met a compiler and is not grown further.
        for li in 0..maps.layers.len() {
            while let Some(e) = self.pop_layer_event_raw(li as u32, 0, maps)? {
                if maps.pending_layer_events.len() <= li {
                    maps.pending_layer_events.resize_with(li + 1, Default::default);
                }
                maps.pending_layer_events[li].push_back(e);
            }
            for instance in 1..n_instances {
                while self.pop_layer_event_raw(li as u32, instance, maps)?.is_some() {}
            }
        }
        check_rc(ctx, unsafe { fyx_machine_clear(ctx, id) })?;
        maps.parameter_index.clear();
        maps.layers.clear();
        self.attach_machine(machine, rig, maps)?;
        for (instance, per_layer) in saved.iter().enumerate() {
            let instance = instance as u32;
            for (li, old) in per_layer.iter().enumerate() {
                let Some(lm) = maps.layers.get(li) else {
                    continue; // the layer is gone
                };
                let li = li as u32;
                let mut s = old.active_state.and_then(|h| lm.state_index.get(&h)).map(|i| *i as i32).unwrap_or(-1);
                if old.active_state.is_none() {
                    // MachineLayer::add_state makes a state active whenever none is (layer.rs:229-235) -- also while a
                    // transition is in flight: the first state this edit ADDED takes the place
                    let added = layer_states_added(machine, li as usize, &old.known_states);
                    if let Some(h) = added {
                        s = lm.state_index.get(&h).map(|i| *i as i32).unwrap_or(-1);
                    }
                }
                let t = old.active_transition.and_then(|h| lm.transition_index.get(&h)).map(|i| *i as i32).unwrap_or(-1);
                check_rc(ctx, unsafe { fyx_layer_set_state(ctx, id, li, instance, s, t) })?;
                for (h, elapsed, factor) in old.transitions.iter() {
                    if let Some(index) = lm.transition_index.get(h) {
                        check_rc(ctx, unsafe { fyx_layer_set_transition_state(ctx, id, li, instance, *index, *elapsed, *factor) })?;
                    }
                }
                for (h, has_prev, prev, time) in old.by_index.iter() {
                    // still a BlendAnimationsByIndex node?  (a freed slot may have been reused by a node of another kind)
                    if lm.by_index_nodes.contains(h) {
                        let node = lm.node_index[h] as u32;
                        check_rc(ctx, unsafe { fyx_layer_set_node_state(ctx, id, li, instance, node, *has_prev, *prev, *time) })?;
                    }
                }
            }
        }
        Ok(())
    }

    /// `MachineLayer::pop_event` (layer.rs:284-286) of instance 0: events kept across a rebuild first, then the library's.
    pub fn pop_layer_event(&mut self, layer: u32, maps: &mut AnimatorMaps) -> Result<Option<Event>, HipError> {
        if let Some(q) = maps.pending_layer_events.get_mut(layer as usize) {
            if let Some(e) = q.pop_front() {
                return Ok(Some(e));
            }
        }
        self.pop_layer_event_raw(layer, 0, maps)
    }

    /// One `fyx_layer_pop_event`, its indices turned back into the handles of `maps` (machine/event.rs:33-51).
    fn pop_layer_event_raw(&mut self, layer: u32, instance: u32, maps: &AnimatorMaps) -> Result<Option<Event>, HipError> {
        let mut ev = FyxLayerEvent { kind: 0, a: -1, b: -1 };
        let mut has = 0i32;
        check_rc(self.raw(), unsafe { fyx_layer_pop_event(self.raw(), self.id(), layer, instance, &mut ev, &mut has) })?;
        if has == 0 {
            return Ok(None);
        }
        let Some(lm) = maps.layers.get(layer as usize) else {
            return Ok(None);
        };
        let state = |i: i32| lm.state_index.iter().find(|(_, k)| **k as i32 == i).map(|(h, _)| *h).unwrap_or_default();
        let transition = |i: i32| lm.transition_index.iter().find(|(_, k)| **k as i32 == i).map(|(h, _)| *h).unwrap_or_default();
        Ok(Some(match ev.kind {
            FYX_EVENT_STATE_ENTER => Event::StateEnter(state(ev.a)),
            FYX_EVENT_STATE_LEAVE => Event::StateLeave(state(ev.a)),
            FYX_EVENT_ACTIVE_STATE_CHANGED => Event::ActiveStateChanged { prev: state(ev.a), new: state(ev.b) },
            _ => Event::ActiveTransitionChanged(transition(ev.a)),
        }))
    }

    /// `MachineLayer::reset` (layer.rs:288-296) for every instance.
    pub fn reset_layer(&mut self, layer: u32) -> Result<(), HipError> {
        check_rc(self.raw(), unsafe { fyx_layer_reset(self.raw(), self.id(), layer, FYX_ALL_INSTANCES) })
    }

    /// Game code changes parameters between frames (`machine.parameters_mut().get_mut(name)`): push the current values
    /// before `update_machine`.  One small call per parameter the graph uses; the values are host scalars.
    pub fn sync_parameters(&mut self, machine: &Machine, maps: &AnimatorMaps) -> Result<(), HipError> {
        for (name, index) in maps.parameter_index.iter() {
            if let Some(p) = machine.parameters().get(name) {
                let (kind, f0, f1, u) = parameter_words(p);
                check_rc(self.raw(), unsafe { fyx_machine_set_parameter(self.raw(), self.id(), *index, FYX_ALL_INSTANCES, kind, f0, f1, u) })?;
            }
        }
        Ok(())
    }
}

/// A hash of everything `attach_machine` sends: when it differs from the one taken at the last attach, the game has edited
/// the `Machine` in place and `sync_machine` re-sends it.  Walks the pools once (a few hundred nanoseconds for a machine
/// of a dozen nodes); run-time fields (active state, elapsed times, cached poses) and parameter VALUES are not part of it.
pub fn machine_signature(machine: &Machine) -> u64 {
    use std::hash::{Hash, Hasher};
    let mut k = std::collections::hash_map::DefaultHasher::new();
    fn logic(node: &LogicNode, k: &mut std::collections::hash_map::DefaultHasher) {
        match node {
            LogicNode::Parameter(name) => {
                0u8.hash(k);
                name.hash(k);
            }
            LogicNode::And(n) => {
                1u8.hash(k);
                logic(&n.lhs, k);
                logic(&n.rhs, k);
            }
            LogicNode::Or(n) => {
                2u8.hash(k);
                logic(&n.lhs, k);
                logic(&n.rhs, k);
            }
            LogicNode::Xor(n) => {
                3u8.hash(k);
                logic(&n.lhs, k);
                logic(&n.rhs, k);
            }
            LogicNode::Not(n) => {
                4u8.hash(k);
                logic(&n.lhs, k);
            }
            LogicNode::IsAnimationEnded(h) => {
                5u8.hash(k);
                h.hash(k);
            }
        }
    }
    machine.layers().len().hash(&mut k);
    for layer in machine.layers() {
        layer.weight().to_bits().hash(&mut k);
        for h in layer.mask().inner().iter() {
            h.hash(&mut k);
        }
        layer.entry_state().hash(&mut k);
        for (h, node) in layer.nodes().pair_iter() {
            h.hash(&mut k);
            match node {
                PoseNode::PlayAnimation(play) => {
                    0u8.hash(&mut k);
                    play.animation.hash(&mut k);
                }
                PoseNode::BlendAnimations(blend) => {
                    1u8.hash(&mut k);
                    for p in blend.pose_sources.iter() {
                        p.pose_source.hash(&mut k);
                        match &p.weight {
                            PoseWeight::Constant(w) => w.to_bits().hash(&mut k),
                            PoseWeight::Parameter(name) => name.hash(&mut k),
                        }
                    }
                }
                PoseNode::BlendAnimationsByIndex(by_index) => {
                    2u8.hash(&mut k);
                    by_index.index_parameter.hash(&mut k);
                    for i in by_index.inputs.iter() {
                        i.pose_source.hash(&mut k);
                        i.blend_time.to_bits().hash(&mut k);
                    }
                }
                PoseNode::BlendSpace(space) => {
                    3u8.hash(&mut k);
                    space.sampling_parameter().hash(&mut k);
                    for p in space.points() {
                        p.position.x.to_bits().hash(&mut k);
                        p.position.y.to_bits().hash(&mut k);
                        p.pose_source.hash(&mut k);
                    }
                }
            }
        }
        for (h, state) in layer.states().pair_iter() {
            h.hash(&mut k);
            state.root.hash(&mut k);
            for actions in [&state.on_enter_actions, &state.on_leave_actions] {
                actions.len().hash(&mut k);
                for action in actions.iter() {
                    match &action.0 {
                        StateAction::None => 0u8.hash(&mut k),
                        StateAction::RewindAnimation(a) => {
                            1u8.hash(&mut k);
                            a.hash(&mut k);
                        }
                        StateAction::EnableAnimation(a) => {
                            2u8.hash(&mut k);
                            a.hash(&mut k);
                        }
                        StateAction::DisableAnimation(a) => {
                            3u8.hash(&mut k);
                            a.hash(&mut k);
                        }
                        StateAction::EnableRandomAnimation(list) => {
                            4u8.hash(&mut k);
                            for a in list.iter() {
                                a.hash(&mut k);
                            }
                        }
                    }
                }
            }
        }
        for (h, transition) in layer.transitions().pair_iter() {
            h.hash(&mut k);
            transition.source().hash(&mut k);
            transition.dest().hash(&mut k);
            transition.transition_time().to_bits().hash(&mut k);
            logic(transition.condition(), &mut k);
        }
    }
    k.finish()
}

/// The first state (in pool order) of layer `li` that the shim did not know before the edit.
fn layer_states_added(machine: &Machine, li: usize, known: &[Handle<State>]) -> Option<Handle<State>> {
    let layer = machine.layers().get(li)?;
    layer.states().pair_iter().map(|(h, _)| h).find(|h| !known.contains(h))
}

/// `Parameter` (machine/parameter.rs:36-50) as the (kind, f0, f1, u) words of `fyx_machine_add_parameter`.
fn parameter_words(p: &Parameter) -> (i32, f32, f32, u32) {
    match p {
        Parameter::Weight(w) => (FYX_PARAM_WEIGHT, *w, 0.0, 0),
        Parameter::Rule(r) => (FYX_PARAM_RULE, 0.0, 0.0, *r as u32),
        Parameter::Index(i) => (FYX_PARAM_INDEX, 0.0, 0.0, *i),
        Parameter::SamplingPoint(v) => (FYX_PARAM_SAMPLING_POINT, v.x, v.y, 0),
    }
}

/// ParameterContainer has no public iteration (machine/parameter.rs:136-200: `add` / `get` / `get_mut` by name), and the
/// library does not need one: every name the graph refers to is resolved through `get` and registered once.  -1 for a
/// name that does not exist: it behaves like a missing parameter in the reference.
fn resolve_parameter(ctx: *mut FyxCtx, id: u64, machine: &Machine, maps: &mut AnimatorMaps, name: &str) -> Result<i32, HipError> {
    if let Some(i) = maps.parameter_index.get(name) {
        return Ok(*i as i32);
    }
    let Some(p) = machine.parameters().get(name) else {
        return Ok(-1);
    };
    let (kind, f0, f1, u) = parameter_words(p);
    let mut index = 0u32;
    check_rc(ctx, unsafe { fyx_machine_add_parameter(ctx, id, kind, f0, f1, u, &mut index) })?;
    maps.parameter_index.insert(name.to_string(), index);
    Ok(index as i32)
}

/// `LogicNode` (machine/transition.rs:118-131) -> the prefix code of `fyx_layer_add_transition`.
fn encode_logic(node: &LogicNode, out: &mut Vec<i32>, ctx: *mut FyxCtx, id: u64, machine: &Machine, maps: &mut AnimatorMaps) -> Result<(), HipError> {
    match node {
        LogicNode::Parameter(name) => {
            out.push(FYX_LOGIC_PARAMETER);
            out.push(resolve_parameter(ctx, id, machine, maps, name)?);
        }
        LogicNode::And(n) => {
            out.push(FYX_LOGIC_AND);
            encode_logic(&n.lhs, out, ctx, id, machine, maps)?;
            encode_logic(&n.rhs, out, ctx, id, machine, maps)?;
        }
        LogicNode::Or(n) => {
            out.push(FYX_LOGIC_OR);
            encode_logic(&n.lhs, out, ctx, id, machine, maps)?;
            encode_logic(&n.rhs, out, ctx, id, machine, maps)?;
        }
        LogicNode::Xor(n) => {
            out.push(FYX_LOGIC_XOR);
            encode_logic(&n.lhs, out, ctx, id, machine, maps)?;
            encode_logic(&n.rhs, out, ctx, id, machine, maps)?;
        }
        LogicNode::Not(n) => {
            out.push(FYX_LOGIC_NOT);
            encode_logic(&n.lhs, out, ctx, id, machine, maps)?;
        }
        LogicNode::IsAnimationEnded(h) => {
            out.push(FYX_LOGIC_IS_ANIMATION_ENDED);
            // a handle that does not resolve is "ended" (transition.rs:167-170): -1 says so to the library
            out.push(maps.animation_index.get(h).map(|a| *a as i32).unwrap_or(-1));
        }
    }
    Ok(())
}
