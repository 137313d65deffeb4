// SPDX-License-Identifier: Apache-2.0
//! Product quantisation on the MI355X engine: encode, per-partition ADC top-k, batched index search.
//! NOT COMPILED in the repository that carries this file; see integration/README.md.

use std::ffi::CString;

use arrow_array::{Float32Array, RecordBatch, UInt64Array};
use lance_core::Result;
use lance_linalg::distance::DistanceType;
use lance_linalg::hip::{self, check, DeviceBuffer, HipDType, LanceHipIndex, HIP_CTX};

/// `ProductQuantizer::transform_impl::<8, f32>` (pq.rs:116-191): L2-nearest codeword per sub-vector (the quantizer's own
/// distance type is always L2, builder.rs:456), `unwrap_or(0)` for all-NaN sub-vectors (pq.rs:165).
pub fn pq_encode_f32(vectors: &[f32], dimension: usize, codebook: &[f32], num_sub_vectors: usize, num_bits: u32) -> Result<Vec<u8>> {
    HIP_CTX.with(|ctx| {
        let n = vectors.len() / dimension;
        let code_bytes = if num_bits == 4 { num_sub_vectors / 2 } else { num_sub_vectors };
        let x = ctx.upload(vectors)?;
        let cb = ctx.upload(codebook)?;
        let codes = DeviceBuffer::<u8>::alloc(ctx, n * code_bytes)?;
        check(unsafe {
            hip::lance_hip_pq_encode(ctx.as_ptr(), HipDType::F32 as i32, 0, x.ptr, n as u64, dimension as u32, cb.ptr,
                                     num_sub_vectors as u32, num_bits, codes.ptr as *mut u8)
        })?;
        codes.download()
    })
}

/// `PQDistCalculator::new` + `distance_all` + `FlatIndex::search` for ONE partition (pq/storage.rs:854-960,
/// flat/index.rs:82-177): residual query, transposed codes `[M][n_p]`, row ids, `k`, optional `[lower, upper)`.
/// Returns the `(_distance, _rowid)` batch `IvfSubIndex::search` returns (v3/subindex.rs:18-61, schema flat/index.rs:41-47).
#[allow(clippy::too_many_arguments)]
pub fn pq_partition_topk(
    q_residual: &[f32],
    codebook: &[f32],
    num_sub_vectors: usize,
    num_bits: u32,
    codes_transposed: &[u8],
    row_ids: &[u64],
    k: usize,
    distance_type: DistanceType,
    range: Option<(f32, f32)>,
) -> Result<RecordBatch> {
    HIP_CTX.with(|ctx| {
        let n_p = row_ids.len();
        let q = ctx.upload(q_residual)?;
        let cb = ctx.upload(codebook)?;
        let codes = ctx.upload(codes_transposed)?;
        let rid = ctx.upload(row_ids)?;
        let out_ids = DeviceBuffer::<u64>::alloc(ctx, k)?;
        let out_d = DeviceBuffer::<f32>::alloc(ctx, k)?;
        let mut got = 0u32;
        let (has, lo, hi) = range.map_or((0, 0.0, 0.0), |(l, u)| (1, l, u));
        check(unsafe {
            hip::lance_hip_pq_scan_topk(ctx.as_ptr(), HipDType::F32 as i32, hip::metric_code(distance_type)?, q.ptr,
                                        q_residual.len() as u32, cb.ptr, num_sub_vectors as u32, num_bits, codes.ptr as *const u8,
                                        rid.ptr as *const u64, n_p as u64, k as u32, has, lo, hi, out_ids.ptr as *mut u64,
                                        out_d.ptr as *mut f32, &mut got)
        })?;
        let (mut ids, mut d) = (out_ids.download()?, out_d.download()?);
        ids.truncate(got as usize);
        d.truncate(got as usize);
        Ok(RecordBatch::try_new(
            super::flat::index::ANN_SEARCH_SCHEMA.clone(),
            vec![std::sync::Arc::new(Float32Array::from(d)), std::sync::Arc::new(UInt64Array::from(ids))],
        )?)
    })
}

/// An IVF_PQ index resident in HBM: opened once per `IVFIndex` (v2.rs:106-200), searched per query BATCH instead of the
/// per-(query, partition) task fan-out of `ANNIvfSubIndexExec` (knn.rs:579-1075).
pub struct HipIvfPqIndex {
    raw: *mut LanceHipIndex,
}
unsafe impl Send for HipIvfPqIndex {}
unsafe impl Sync for HipIvfPqIndex {}

impl HipIvfPqIndex {
    /// `<dataset>/_indices/<uuid>` holding `index.idx` + `auxiliary.idx` (local file systems; object stores keep the Rust
    /// reader and call `lance_hip_index_from_storage` with the arrays).
    pub fn open(index_dir: &str, dtype: HipDType) -> Result<Self> {
        HIP_CTX.with(|ctx| {
            let dir = CString::new(index_dir).unwrap();
            let mut raw = std::ptr::null_mut();
            check(unsafe { hip::lance_hip_index_load(ctx.as_ptr(), dir.as_ptr(), dtype as i32, &mut raw) })?;
            Ok(Self { raw })
        })
    }

    /// `nearest = {q, k, nprobes, refine_factor}` for a batch; `prefilter`: one byte per row id (`DatasetPreFilter` mask).
    pub fn search(&self, queries: &[f32], dimension: usize, k: usize, nprobes: usize, refine_factor: Option<u32>,
                  prefilter: Option<&[u8]>) -> Result<(Vec<u64>, Vec<f32>)> {
        HIP_CTX.with(|ctx| {
            let nq = queries.len() / dimension;
            let q = ctx.upload(queries)?;
            let ids = DeviceBuffer::<u64>::alloc(ctx, nq * k)?;
            let d = DeviceBuffer::<f32>::alloc(ctx, nq * k)?;
            let rf = refine_factor.unwrap_or(0);
            match prefilter {
                None => check(unsafe {
                    hip::lance_hip_ivfpq_search(ctx.as_ptr(), self.raw, q.ptr, nq as u32, k as u32, nprobes as u32, rf,
                                                ids.ptr as *mut u64, d.ptr as *mut f32)
                })?,
                Some(mask) => {
                    let m = ctx.upload(mask)?;
                    check(unsafe {
                        hip::lance_hip_ivfpq_search_filtered(ctx.as_ptr(), self.raw, q.ptr, nq as u32, k as u32, nprobes as u32, rf,
                                                             m.ptr as *const u8, mask.len() as u64, ids.ptr as *mut u64, d.ptr as *mut f32)
                    })?
                }
            }
            Ok((ids.download()?, d.download()?))
        })
    }
}
impl Drop for HipIvfPqIndex {
    fn drop(&mut self) {
        unsafe { hip::lance_hip_index_destroy(self.raw) }
    }
}
