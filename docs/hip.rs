// docs/hip.rs -- the Rust side of the drop-in boundary: `rust/src/collector/hip.rs` as a maintainer of AI4quantum/twisteRL
// would add it, complete (every struct of include/twisterl_hip.h that the path uses, all eight result fields, both
// collectors, the `Box<dyn Env>` thunks).
//
// *** NOT COMPILED, NOT TESTED: this image has no Rust toolchain (rustc / cargo absent, no network).  The C ABI it binds is
// *** exercised from compiled C (examples/collect_from_c.c) and from Python (twisterl_amd/_lib.py, the whole test suite);
// *** struct layouts below are field-for-field copies of include/twisterl_hip.h at ABI version 6 (6: tw_env_vtable grew the rest of the trait).
//
// What else the reference needs (three small additions, all `pub(crate)`):
//   rust/build.rs                 println!("cargo:rustc-link-search=native={}", env::var("TWISTERL_HIP_LIB_DIR").unwrap());
//                                 println!("cargo:rustc-link-lib=dylib=twisterl_hip");
//   rust/src/nn/layers.rs         impl Linear       { pub(crate) fn parts(&self) -> (&[f32], &[f32], bool) { (self.weights.as_slice(), self.bias.as_slice(), self.apply_relu) } }
//                                 impl EmbeddingBag { pub(crate) fn parts(&self) -> (&Vec<DVector<f32>>, &[f32], bool, &Vec<usize>, usize) { (&self.vectors, self.bias.as_slice(), self.apply_relu, &self.obs_shape, self.conv_dim) } }
//                                 (DMatrix::from_vec(out, in, data) is column-major: as_slice() IS `data`, i.e. [in][out] row-major = torch_weight.T.flatten(), layers.rs:26)
//   rust/src/nn/modules.rs        impl Sequential   { pub(crate) fn layers(&self) -> &Vec<Box<Linear>> { &self.layers } }
//   rust/src/nn/policy.rs         impl Policy       { pub(crate) fn parts(&self) -> (&EmbeddingBag, &Sequential, &Sequential, &Sequential, &Vec<Vec<usize>>, &Vec<Vec<usize>>) { .. } }
//   rust/src/python_interface/collector.rs:159-187   PyPPOCollector::new / PyAZCollector::new box a Hip*Collector when
//                                 `tw_device_count() > 0`, else the rayon one; nothing above PyBaseCollector::collect changes.

use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

use anyhow::{anyhow, Result};

use crate::collector::collector::{CollectedData, Collector};
use crate::envs::puzzle::Puzzle;
use crate::nn::policy::Policy;
use crate::rl::env::Env;

// ---------------------------------------------------------------------------------------------- include/twisterl_hip.h
#[repr(C)] pub struct TwPuzzleDesc { width: u32, height: u32, difficulty: u32, depth_slope: u32, max_depth: u32 }

#[repr(C)] pub struct TwLinearDesc { in_features: u32, out_features: u32, weights: *const f32, bias: *const f32, apply_relu: u32 }

#[repr(C)] pub struct TwPolicyDesc {
    obs_size: u32, emb_size: u32,
    emb_vectors: *const f32,          // [obs_size][emb_size] == torch_weight.T
    emb_bias: *const f32,             // [emb_size]
    emb_apply_relu: u32,
    n_common: u32, common: *const TwLinearDesc,
    n_action: u32, action: *const TwLinearDesc,
    n_value: u32,  value: *const TwLinearDesc,
    n_perms: u32, n_actions: u32,
    obs_perms: *const i32,            // [n_perms][obs_size]
    act_perms: *const i32,            // [n_perms][n_actions]
}

#[repr(C)] pub struct TwPpoParams { num_episodes: u64, episode_offset: u64, gamma: f32, lambda: f32, seed: u64,
                                    precision: u32, merge_order: u32, reserve_cus: u32 }
#[repr(C)] pub struct TwAzParams  { num_episodes: u64, episode_offset: u64, num_mcts_searches: u32, c: f32, max_expand_depth: u32,
                                    seed: u64, precision: u32, merge_order: u32, reserve_cus: u32 }

/// `trait Env` as a table of C functions (tw_env_vtable): what tw_ppo_collect_env / tw_az_collect_env call for an environment
/// whose dynamics the kernels do not implement (GridWorld-style crates, examples/grid_world/src/lib.rs:84-164).
#[repr(C)] pub struct TwEnvVtable {
    prototype: *mut c_void,
    num_actions: u32, n_obs: u32, obs_size: u32,
    clone: extern "C" fn(*mut c_void) -> *mut c_void,
    destroy: extern "C" fn(*mut c_void),
    reset: extern "C" fn(*mut c_void, u64, u64),
    step: extern "C" fn(*mut c_void, u32),
    observe: extern "C" fn(*mut c_void, *mut i32),
    masks: extern "C" fn(*mut c_void, *mut u8),
    reward: extern "C" fn(*mut c_void) -> f32,
    is_final: extern "C" fn(*mut c_void) -> c_int,
    success: Option<extern "C" fn(*mut c_void) -> c_int>,
    // the rest of the trait (rl/env.rs:30,58-66); None = the trait's default body (ABI 6)
    track_solution: Option<extern "C" fn(*mut c_void) -> c_int>,
    solution: Option<extern "C" fn(*mut c_void, *mut u32, u32) -> u32>,
    set_state: Option<extern "C" fn(*mut c_void, *const i64, u32)>,
    twists: Option<extern "C" fn(*mut c_void, *mut i32, *mut i32, u32) -> u32>,
}

// fields of the result (TW_F_*)
const F_OBS: c_int = 0; const F_LOGITS: c_int = 1; const F_PERMS: c_int = 2; const F_VALUES: c_int = 3; const F_REWARDS: c_int = 4;
const F_ACTIONS: c_int = 5; const F_ADVS: c_int = 6; const F_RETS: c_int = 7; const F_REMAINING: c_int = 8;

extern "C" {
    fn tw_last_error() -> *const c_char;
    fn tw_device_count() -> c_int;
    fn tw_policy_create(desc: *const TwPolicyDesc) -> *mut c_void;
    fn tw_policy_destroy(p: *mut c_void);
    fn tw_ppo_collect(env: *const TwPuzzleDesc, policy: *const c_void, prm: *const TwPpoParams, out: *mut *mut c_void) -> c_int;
    fn tw_az_collect(env: *const TwPuzzleDesc, policy: *const c_void, prm: *const TwAzParams, out: *mut *mut c_void) -> c_int;
    fn tw_ppo_collect_env(env: *const TwEnvVtable, policy: *const c_void, prm: *const TwPpoParams, max_records: u32, out: *mut *mut c_void) -> c_int;
    fn tw_az_collect_env(env: *const TwEnvVtable, policy: *const c_void, prm: *const TwAzParams, max_records: u32, out: *mut *mut c_void) -> c_int;
    fn tw_collected_num_records(c: *const c_void) -> u64;
    fn tw_collected_num_cells(c: *const c_void) -> u32;
    fn tw_collected_num_actions(c: *const c_void) -> u32;
    fn tw_collected_obs_width(c: *const c_void) -> u32;
    fn tw_collected_copy_to_host(c: *const c_void, field: c_int, dst: *mut c_void, bytes: usize) -> c_int;
    fn tw_collected_free(c: *mut c_void);
}

fn last_error() -> anyhow::Error {
    // anyhow::Error -> MyError -> PyRuntimeError, as every error of the collectors (python_interface/error_mapping.rs:20-33)
    anyhow!(unsafe { CStr::from_ptr(tw_last_error()) }.to_string_lossy().into_owned())
}

pub fn hip_available() -> bool { unsafe { tw_device_count() > 0 } }

// ---------------------------------------------------------------------------------------------- policy hand-off
/// The device copy of a `Policy` (tw_policy_create copies the weights; the host vectors need not outlive the call).
/// The reference rebuilds its Rust Policy every iteration (`policy.to_rust()`, src/twisterl/rl/algorithm.py:90-93), so one
/// handle per `collect` is the same cost model; a cache keyed by the Policy's address would avoid even that.
struct HipPolicy(*mut c_void);
impl Drop for HipPolicy { fn drop(&mut self) { unsafe { tw_policy_destroy(self.0) } } }

impl HipPolicy {
    fn new(policy: &Policy) -> Result<Self> {
        let (emb, common, action, value, obs_perms, act_perms) = policy.parts();
        let (vectors, emb_bias, emb_relu, obs_shape, _conv_dim) = emb.parts();
        if obs_shape.len() != 1 {
            // Conv1dPolicy's two-entry obs_shape (layers.rs:63-77): expand to the dense [obs_size][emb] table first, as
            // twisterl_amd.nn.EmbeddingBag.dense_table() does -- omitted here
            return Err(anyhow!("hip: conv1d EmbeddingBag: expand to a dense table before tw_policy_create"));
        }
        let emb_size = emb_bias.len();
        let table: Vec<f32> = vectors.iter().flat_map(|v| v.iter().copied()).collect();      // [obs_size][emb_size]
        let descs = |s: &crate::nn::modules::Sequential| -> Vec<TwLinearDesc> {
            s.layers().iter().map(|l| { let (w, b, relu) = l.parts();
                TwLinearDesc { in_features: (w.len() / b.len()) as u32, out_features: b.len() as u32, weights: w.as_ptr(), bias: b.as_ptr(), apply_relu: relu as u32 } }).collect()
        };
        let (dc, da, dv) = (descs(common), descs(action), descs(value));
        let n_actions = da.last().map(|l| l.out_features).unwrap_or(0);
        let op: Vec<i32> = obs_perms.iter().flat_map(|p| p.iter().map(|&x| x as i32)).collect();
        let ap: Vec<i32> = act_perms.iter().flat_map(|p| p.iter().map(|&x| x as i32)).collect();
        let desc = TwPolicyDesc {
            obs_size: vectors.len() as u32, emb_size: emb_size as u32, emb_vectors: table.as_ptr(), emb_bias: emb_bias.as_ptr(), emb_apply_relu: emb_relu as u32,
            n_common: dc.len() as u32, common: dc.as_ptr(), n_action: da.len() as u32, action: da.as_ptr(), n_value: dv.len() as u32, value: dv.as_ptr(),
            n_perms: obs_perms.len() as u32, n_actions, obs_perms: op.as_ptr(), act_perms: ap.as_ptr(),
        };
        let h = unsafe { tw_policy_create(&desc) };
        if h.is_null() { Err(last_error()) } else { Ok(HipPolicy(h)) }
    }
}

// ---------------------------------------------------------------------------------------------- Box<dyn Env> behind the C table
// The handle the library passes round is a `*mut Box<dyn Env>` (a thin pointer to the fat one).  The collector clones the
// prototype per episode and never mutates it (ppo.rs:59); `reset` gets the collect's seed and the GLOBAL episode index (a
// build extension: the reference's envs draw from thread_rng and ignore them).
extern "C" fn env_clone(e: *mut c_void) -> *mut c_void { let b = unsafe { &*(e as *mut Box<dyn Env>) }; Box::into_raw(Box::new(dyn_clone::clone_box(&**b))) as *mut c_void }
extern "C" fn env_destroy(e: *mut c_void) { drop(unsafe { Box::from_raw(e as *mut Box<dyn Env>) }) }
extern "C" fn env_reset(e: *mut c_void, _seed: u64, _episode: u64) { unsafe { (*(e as *mut Box<dyn Env>)).reset() } }
extern "C" fn env_step(e: *mut c_void, a: u32) { unsafe { (*(e as *mut Box<dyn Env>)).step(a as usize) } }
extern "C" fn env_observe(e: *mut c_void, out: *mut i32) {
    // tw_env_vtable.n_obs ids per state, always (include/twisterl_hip.h): an env whose observe() changes length cannot use this path
    let obs = unsafe { (*(e as *mut Box<dyn Env>)).observe() };
    for (i, &v) in obs.iter().enumerate() { unsafe { *out.add(i) = v as i32 } }
}
extern "C" fn env_masks(e: *mut c_void, out: *mut u8) { let m = unsafe { (*(e as *mut Box<dyn Env>)).masks() }; for (i, &v) in m.iter().enumerate() { unsafe { *out.add(i) = v as u8 } } }
extern "C" fn env_reward(e: *mut c_void) -> f32 { unsafe { (*(e as *mut Box<dyn Env>)).reward() } }
extern "C" fn env_is_final(e: *mut c_void) -> c_int { unsafe { (*(e as *mut Box<dyn Env>)).is_final() as c_int } }
extern "C" fn env_success(e: *mut c_void) -> c_int { unsafe { (*(e as *mut Box<dyn Env>)).success() as c_int } }
// Env::track_solution / Env::solution (rl/env.rs:61-66): tw_solve_env32 asks track_solution once per attempt, before its first move,
// and returns solution() in place of the played actions when it is true -- single_solve (rl/solve.rs:28,57-64)
extern "C" fn env_track_solution(e: *mut c_void) -> c_int { unsafe { (*(e as *mut Box<dyn Env>)).track_solution() as c_int } }
extern "C" fn env_solution(e: *mut c_void, out: *mut u32, cap: u32) -> u32 {
    let sol = unsafe { (*(e as *mut Box<dyn Env>)).solution() };
    for (i, &v) in sol.iter().take(cap as usize).enumerate() { unsafe { *out.add(i) = v as u32 } }
    sol.len() as u32
}
extern "C" fn env_set_state(e: *mut c_void, st: *const i64, n: u32) {
    let v = unsafe { std::slice::from_raw_parts(st, n as usize) }.to_vec();
    unsafe { (*(e as *mut Box<dyn Env>)).set_state(v) }
}
extern "C" fn env_twists(e: *mut c_void, obs_perms: *mut i32, act_perms: *mut i32, cap: u32) -> u32 {
    let b = unsafe { &*(e as *mut Box<dyn Env>) };
    let (op, ap) = b.twists();
    let (obs_size, n_act) = (b.obs_shape().iter().product::<usize>(), b.num_actions());
    for k in 0..op.len().min(cap as usize) {
        for (i, &v) in op[k].iter().enumerate() { unsafe { *obs_perms.add(k * obs_size + i) = v as i32 } }
        for (i, &v) in ap[k].iter().enumerate() { unsafe { *act_perms.add(k * n_act + i) = v as i32 } }
    }
    op.len() as u32
}

fn vtable_of(env: &Box<dyn Env>) -> TwEnvVtable {
    TwEnvVtable {
        prototype: env as *const Box<dyn Env> as *mut c_void,
        num_actions: env.num_actions() as u32, n_obs: env.observe().len() as u32, obs_size: env.obs_shape().iter().product::<usize>() as u32,
        clone: env_clone, destroy: env_destroy, reset: env_reset, step: env_step, observe: env_observe, masks: env_masks,
        reward: env_reward, is_final: env_is_final, success: Some(env_success),
        track_solution: Some(env_track_solution), solution: Some(env_solution), set_state: Some(env_set_state), twists: Some(env_twists),
    }
}

fn puzzle_desc(env: &Box<dyn Env>) -> Option<TwPuzzleDesc> {
    // envs whose dynamics the kernels implement run wholly on the GPU (env.rs:20 as_any; the pattern of python_interface/env.rs:24-29)
    env.as_any().downcast_ref::<Puzzle>().map(|p| TwPuzzleDesc { width: p.width as u32, height: p.height as u32,
        difficulty: p.difficulty as u32, depth_slope: p.depth_slope as u32, max_depth: p.max_depth as u32 })
}

// ---------------------------------------------------------------------------------------------- result -> CollectedData
struct Handle(*mut c_void);
impl Drop for Handle { fn drop(&mut self) { unsafe { tw_collected_free(self.0) } } }

fn fetch<T: Clone + Default>(h: &Handle, field: c_int, n: usize) -> Result<Vec<T>> {
    let mut v = vec![T::default(); n];
    if n > 0 && unsafe { tw_collected_copy_to_host(h.0, field, v.as_mut_ptr() as *mut c_void, n * std::mem::size_of::<T>()) } != 0 { return Err(last_error()); }
    Ok(v)
}

fn obs_and_logits(h: &Handle) -> Result<(Vec<Vec<usize>>, Vec<Vec<f32>>, Vec<Option<usize>>, usize)> {
    let n = unsafe { tw_collected_num_records(h.0) } as usize;
    let nc = unsafe { tw_collected_num_cells(h.0) } as usize;
    let na = unsafe { tw_collected_num_actions(h.0) } as usize;
    let obs: Vec<Vec<usize>> = if unsafe { tw_collected_obs_width(h.0) } == 2 {          // ids beyond 255: two bytes each
        fetch::<u16>(h, F_OBS, n * nc)?.chunks(nc).map(|r| r.iter().map(|&x| x as usize).collect()).collect()
    } else {
        fetch::<u8>(h, F_OBS, n * nc)?.chunks(nc).map(|r| r.iter().map(|&x| x as usize).collect()).collect()
    };
    let logits = fetch::<f32>(h, F_LOGITS, n * na)?.chunks(na).map(|r| r.to_vec()).collect();
    let perms = fetch::<i8>(h, F_PERMS, n)?.into_iter().map(|p| if p < 0 { None } else { Some(p as usize) }).collect();
    Ok((obs, logits, perms, n))
}

// ---------------------------------------------------------------------------------------------- the collectors
#[derive(Clone)]
pub struct HipPPOCollector { pub num_episodes: usize, pub gamma: f32, pub lambda: f32, pub num_cores: usize, pub seed: u64, pub max_records: u32 }

impl Collector for HipPPOCollector {
    fn collect(&self, env: &Box<dyn Env>, policy: &Policy) -> Result<CollectedData> {
        if self.num_episodes == 0 { return Err(anyhow!("Something went wrong. No data in collected data chunks to merge. ")); }   // collector.rs:41
        let pol = HipPolicy::new(policy)?;
        let prm = TwPpoParams { num_episodes: self.num_episodes as u64, episode_offset: 0, gamma: self.gamma, lambda: self.lambda,
                                seed: self.seed, precision: 0 /* TW_PREC_F32_EXACT */, merge_order: 1, reserve_cus: 0 };
        let mut out: *mut c_void = std::ptr::null_mut();
        let rc = match puzzle_desc(env) {
            Some(desc) => unsafe { tw_ppo_collect(&desc, pol.0, &prm, &mut out) },
            None => { let vt = vtable_of(env); unsafe { tw_ppo_collect_env(&vt, pol.0, &prm, self.max_records, &mut out) } }
        };
        if rc != 0 { return Err(last_error()); }
        let h = Handle(out);
        let (obs, logits, perms, n) = obs_and_logits(&h)?;
        let values = fetch::<f32>(&h, F_VALUES, n)?;
        let rewards = fetch::<f32>(&h, F_REWARDS, n)?;
        let actions = fetch::<u8>(&h, F_ACTIONS, n)?.into_iter().map(|a| a as usize).collect();
        let mut data = CollectedData::new(obs, logits, perms, values, rewards, actions);                 // collector.rs:50-57
        data.additional_data.insert("advs".to_string(), fetch::<f32>(&h, F_ADVS, n)?);                  // ppo.rs:102-103
        data.additional_data.insert("rets".to_string(), fetch::<f32>(&h, F_RETS, n)?);
        Ok(data)
    }
}

#[derive(Clone)]
#[allow(non_snake_case)]
pub struct HipAZCollector { pub num_episodes: usize, pub num_mcts_searches: usize, pub C: f32, pub max_expand_depth: usize, pub num_cores: usize,
                            pub seed: u64, pub max_records: u32 }

impl Collector for HipAZCollector {
    fn collect(&self, env: &Box<dyn Env>, policy: &Policy) -> Result<CollectedData> {
        if self.num_episodes == 0 { return Err(anyhow!("Something went wrong. No data in collected data chunks to merge. ")); }
        let pol = HipPolicy::new(policy)?;
        let prm = TwAzParams { num_episodes: self.num_episodes as u64, episode_offset: 0, num_mcts_searches: self.num_mcts_searches as u32, c: self.C,
                               max_expand_depth: self.max_expand_depth as u32, seed: self.seed, precision: 0, merge_order: 1, reserve_cus: 0 };
        let mut out: *mut c_void = std::ptr::null_mut();
        let rc = match puzzle_desc(env) {
            Some(desc) => unsafe { tw_az_collect(&desc, pol.0, &prm, &mut out) },
            None => { let vt = vtable_of(env); unsafe { tw_az_collect_env(&vt, pol.0, &prm, self.max_records, &mut out) } }
        };
        if rc != 0 { return Err(last_error()); }
        let h = Handle(out);
        let (obs, probs, perms, n) = obs_and_logits(&h)?;                     // MCTS probs travel in the `logits` slot, perms all None (az.rs:95-104)
        let mut data = CollectedData::new(obs, probs, perms, vec![], vec![], vec![]);
        let mut extra: HashMap<String, Vec<f32>> = HashMap::new();
        extra.insert("remaining_values".to_string(), fetch::<f32>(&h, F_REMAINING, n)?);               // az.rs:93,105
        data.additional_data = extra;
        Ok(data)
    }
}
