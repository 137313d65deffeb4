// SPDX-License-Identifier: Apache-2.0
//! MI355X (gfx950) engine for the IVF-PQ hot path: FFI to `liblance_hip.so` (C ABI: include/lance_hip.h).
//!
//! NOT COMPILED in the repository that carries this file (no Rust toolchain there); written against
//! lance-linalg at the surveyed commit.  Enable with `--features hip`; `build.rs` adds
//! `println!("cargo:rustc-link-lib=dylib=lance_hip");` next to the existing `cc` kernels (build.rs:43-132).

use std::ffi::{c_char, c_void, CStr};
use std::marker::PhantomData;
use std::ptr;

use lance_core::{Error, Result};
use snafu::location;

#[repr(C)]
pub struct LanceHipCtx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct LanceHipIndex {
    _p: [u8; 0],
}

/// `enum { LANCE_HIP_F32 = 0, LANCE_HIP_F16 = 1, LANCE_HIP_I8 = 2 }`
#[repr(i32)]
#[derive(Copy, Clone, Debug, PartialEq, Eq)]
pub enum HipDType {
    F32 = 0,
    F16 = 1,
    I8 = 2,
}

/// `enum { LANCE_HIP_L2 = 0, LANCE_HIP_COSINE = 1, LANCE_HIP_DOT = 2 }`
pub fn metric_code(dt: crate::distance::DistanceType) -> Result<i32> {
    use crate::distance::DistanceType::*;
    match dt {
        L2 => Ok(0),
        Cosine => Ok(1),
        Dot => Ok(2),
        other => Err(Error::Index {
            message: format!("distance type {other} is not on the accelerated path"),
            location: location!(),
        }),
    }
}

#[link(name = "lance_hip")]
extern "C" {
    pub fn lance_hip_ctx_create(device_id: i32, stream: *mut c_void, out: *mut *mut LanceHipCtx) -> i32;
    pub fn lance_hip_ctx_destroy(ctx: *mut LanceHipCtx);
    pub fn lance_hip_last_error() -> *const c_char;
    pub fn lance_hip_synchronize(ctx: *mut LanceHipCtx) -> i32;
    pub fn lance_hip_malloc(ctx: *mut LanceHipCtx, bytes: usize, out: *mut *mut c_void) -> i32;
    pub fn lance_hip_free(ctx: *mut LanceHipCtx, p: *mut c_void) -> i32;
    pub fn lance_hip_memcpy_h2d(ctx: *mut LanceHipCtx, dst: *mut c_void, src: *const c_void, bytes: usize) -> i32;
    pub fn lance_hip_memcpy_d2h(ctx: *mut LanceHipCtx, dst: *mut c_void, src: *const c_void, bytes: usize) -> i32;

    pub fn lance_hip_assign(ctx: *mut LanceHipCtx, dtype: i32, metric: i32, x: *const c_void, n: u64, d: u32,
        centroids: *const c_void, k: u32, bias: *const f32, ids: *mut u32, dists: *mut f32) -> i32;
    pub fn lance_hip_kmeans_train_ex(ctx: *mut LanceHipCtx, dtype: i32, metric: i32, x: *const c_void, n: u64, d: u32, k: u32,
        max_iters: u32, tol: f64, balance_factor: f32, hierarchical_k: u32, init: *const c_void, seed: u64,
        centroids_out: *mut c_void, loss_out: *mut f64, iters_out: *mut u32, k_out: *mut u32) -> i32;
    pub fn lance_hip_pq_train(ctx: *mut LanceHipCtx, dtype: i32, residuals: *const c_void, n: u64, d: u32, m: u32,
        nbits: u32, max_iters: u32, sample_rate: u32, seed: u64, codebook_out: *mut c_void, iters_out: *mut u32) -> i32;
    pub fn lance_hip_pq_encode(ctx: *mut LanceHipCtx, dtype: i32, metric: i32, x: *const c_void, n: u64, d: u32,
        codebook: *const c_void, m: u32, nbits: u32, codes: *mut u8) -> i32;
    pub fn lance_hip_ivfpq_encode(ctx: *mut LanceHipCtx, dtype: i32, metric: i32, x: *const c_void, n: u64, d: u32,
        centroids: *const c_void, nlist: u32, codebook: *const c_void, m: u32, nbits: u32,
        part_ids: *mut u32, codes: *mut u8, loss_out: *mut f64) -> i32;
    pub fn lance_hip_find_partitions(ctx: *mut LanceHipCtx, dtype: i32, metric: i32, q: *const c_void, nq: u32, d: u32,
        centroids: *const c_void, nlist: u32, nprobes: u32, part_ids: *mut u32, dists: *mut f32) -> i32;
    pub fn lance_hip_pq_scan_topk(ctx: *mut LanceHipCtx, dtype: i32, metric: i32, q_residual: *const c_void, d: u32,
        codebook: *const c_void, m: u32, nbits: u32, codes_transposed: *const u8, row_ids: *const u64, n_p: u64,
        k: u32, has_range: i32, lower: f32, upper: f32, out_ids: *mut u64, out_dists: *mut f32, out_n: *mut u32) -> i32;
    pub fn lance_hip_index_from_storage(ctx: *mut LanceHipCtx, dtype: i32, metric: i32, d: u32, centroids: *const c_void,
        nlist: u32, codebook: *const c_void, m: u32, nbits: u32, part_offsets_host: *const u32, codes: *const u8,
        transposed: i32, row_ids: *const u64, n: u64, out: *mut *mut LanceHipIndex) -> i32;
    pub fn lance_hip_index_load(ctx: *mut LanceHipCtx, dir: *const c_char, dtype: i32, out: *mut *mut LanceHipIndex) -> i32;
    pub fn lance_hip_index_set_raw(idx: *mut LanceHipIndex, x: *const c_void, n_raw: u64) -> i32;
    /// Index::prewarm (lance/src/index/vector/ivf/v2.rs:349): build the per-index search constants now.
    pub fn lance_hip_index_prewarm(ctx: *mut LanceHipCtx, idx: *mut LanceHipIndex) -> i32;
    pub fn lance_hip_index_destroy(idx: *mut LanceHipIndex);
    pub fn lance_hip_ivfpq_search(ctx: *mut LanceHipCtx, idx: *const LanceHipIndex, q: *const c_void, nq: u32, k: u32,
        nprobes: u32, refine_factor: u32, ids: *mut u64, dists: *mut f32) -> i32;
    pub fn lance_hip_ivfpq_search_filtered(ctx: *mut LanceHipCtx, idx: *const LanceHipIndex, q: *const c_void, nq: u32, k: u32,
        nprobes: u32, refine_factor: u32, allow_by_rowid: *const u8, n_allow: u64, ids: *mut u64, dists: *mut f32) -> i32;
    pub fn lance_hip_ivfpq_search_range(ctx: *mut LanceHipCtx, idx: *const LanceHipIndex, q: *const c_void, nq: u32, k: u32,
        nprobes: u32, refine_factor: u32, lower: f32, upper: f32, ids: *mut u64, dists: *mut f32) -> i32;
    // PreFilter::mask with Query::lower_bound / upper_bound (flat/index.rs:131-149): both tested inside the scan kernels
    pub fn lance_hip_ivfpq_search_filtered_range(ctx: *mut LanceHipCtx, idx: *const LanceHipIndex, q: *const c_void, nq: u32, k: u32,
        nprobes: u32, refine_factor: u32, allow_by_rowid: *const u8, n_allow: u64, lower: f32, upper: f32,
        ids: *mut u64, dists: *mut f32) -> i32;
    // IVF_FLAT sub-index (FlatIndex over FlatFloatStorage, flat/index.rs:82-177), with and without a RowIdMask
    pub fn lance_hip_ivfflat_create(ctx: *mut LanceHipCtx, dtype: i32, metric: i32, d: u32, centroids: *const c_void, nlist: u32,
        x: *const c_void, part_ids: *const u32, row_ids: *const u64, n: u64, out: *mut *mut LanceHipIndex) -> i32;
    pub fn lance_hip_ivfflat_search(ctx: *mut LanceHipCtx, idx: *const LanceHipIndex, q: *const c_void, nq: u32, k: u32, nprobes: u32,
        ids: *mut u64, dists: *mut f32) -> i32;
    pub fn lance_hip_ivfflat_search_filtered(ctx: *mut LanceHipCtx, idx: *const LanceHipIndex, q: *const c_void, nq: u32, k: u32,
        nprobes: u32, allow_by_rowid: *const u8, n_allow: u64, ids: *mut u64, dists: *mut f32) -> i32;
    pub fn lance_hip_flat_topk(ctx: *mut LanceHipCtx, dtype: i32, metric: i32, x: *const c_void, row_ids: *const u64,
        n: u64, d: u32, q: *const c_void, nq: u32, k: u32, ids: *mut u64, dists: *mut f32) -> i32;
}

/// Maps a return code to the error type the reference path already produces (`Error::Index`), with the library's message.
pub fn check(rc: i32) -> Result<()> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(lance_hip_last_error()) }.to_string_lossy().into_owned();
    Err(Error::Index { message: msg, location: location!() })
}

/// One stream + scratch arena.  Used by one thread at a time: keep one per tokio blocking thread (thread_local below), the way
/// `spawn_cpu` tasks call the CPU kernels today.
pub struct HipContext {
    raw: *mut LanceHipCtx,
}
unsafe impl Send for HipContext {}

impl HipContext {
    pub fn new(device_id: i32) -> Result<Self> {
        let mut raw = ptr::null_mut();
        check(unsafe { lance_hip_ctx_create(device_id, ptr::null_mut(), &mut raw) })?;
        Ok(Self { raw })
    }
    pub fn as_ptr(&self) -> *mut LanceHipCtx {
        self.raw
    }
    /// Host slice -> device buffer (pinned staging inside the library).
    pub fn upload<T: Copy>(&self, host: &[T]) -> Result<DeviceBuffer<'_, T>> {
        let buf = DeviceBuffer::<T>::alloc(self, host.len())?;
        check(unsafe {
            lance_hip_memcpy_h2d(self.raw, buf.ptr, host.as_ptr() as *const c_void, std::mem::size_of_val(host))
        })?;
        Ok(buf)
    }
}
impl Drop for HipContext {
    fn drop(&mut self) {
        unsafe { lance_hip_ctx_destroy(self.raw) }
    }
}

thread_local! {
    /// `LANCE_HIP_DEVICE` selects the GPU (one process per GPU in multi-GPU deployments).
    pub static HIP_CTX: HipContext = HipContext::new(
        std::env::var("LANCE_HIP_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0)
    ).expect("lance_hip context");
}

pub struct DeviceBuffer<'a, T> {
    ctx: &'a HipContext,
    pub ptr: *mut c_void,
    pub len: usize,
    _t: PhantomData<T>,
}
impl<'a, T: Copy> DeviceBuffer<'a, T> {
    pub fn alloc(ctx: &'a HipContext, len: usize) -> Result<Self> {
        let mut p = ptr::null_mut();
        check(unsafe { lance_hip_malloc(ctx.raw, (len.max(1)) * std::mem::size_of::<T>(), &mut p) })?;
        Ok(Self { ctx, ptr: p, len, _t: PhantomData })
    }
    pub fn download(&self) -> Result<Vec<T>> {
        let mut out = Vec::<T>::with_capacity(self.len);
        check(unsafe {
            lance_hip_memcpy_d2h(self.ctx.raw, out.as_mut_ptr() as *mut c_void, self.ptr, self.len * std::mem::size_of::<T>())
        })?;
        unsafe { out.set_len(self.len) };
        Ok(out)
    }
}
impl<T> Drop for DeviceBuffer<'_, T> {
    fn drop(&mut self) {
        unsafe { lance_hip_free(self.ctx.raw, self.ptr) };
    }
}

/// `compute_partitions` / `KMeansAlgoFloat::compute_membership_and_dist` on the device: ids (`None` = 0xFFFF_FFFF) and
/// un-biased distances.  `bias` = `balance_factor * cluster_sizes` (kmeans.rs:345-352).
pub fn assign_f32(
    data: &[f32],
    dimension: usize,
    centroids: &[f32],
    metric: crate::distance::DistanceType,
    bias: Option<&[f32]>,
) -> Result<(Vec<Option<u32>>, Vec<Option<f32>>)> {
    HIP_CTX.with(|ctx| {
        let n = data.len() / dimension;
        let k = centroids.len() / dimension;
        let x = ctx.upload(data)?;
        let c = ctx.upload(centroids)?;
        let b = bias.map(|b| ctx.upload(b)).transpose()?;
        let ids = DeviceBuffer::<u32>::alloc(ctx, n)?;
        let dists = DeviceBuffer::<f32>::alloc(ctx, n)?;
        check(unsafe {
            lance_hip_assign(
                ctx.as_ptr(), HipDType::F32 as i32, metric_code(metric)?, x.ptr, n as u64, dimension as u32, c.ptr, k as u32,
                b.as_ref().map_or(ptr::null(), |b| b.ptr as *const f32), ids.ptr as *mut u32, dists.ptr as *mut f32,
            )
        })?;
        let (ids, dists) = (ids.download()?, dists.download()?);
        Ok(ids
            .iter()
            .zip(dists.iter())
            .map(|(&i, &d)| if i == u32::MAX { (None, None) } else { (Some(i), Some(d)) })
            .unzip())
    })
}
