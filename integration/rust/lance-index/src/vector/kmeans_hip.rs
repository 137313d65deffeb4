// SPDX-License-Identifier: Apache-2.0
//! k-means on the MI355X engine: `KMeansAlgo` (kmeans.rs:239-303) and the whole-training shortcut.
//! NOT COMPILED in the repository that carries this file; see integration/README.md.

use std::sync::Arc;

use arrow_array::cast::AsArray;
use arrow_array::types::Float32Type;
use arrow_array::{Array, Float32Array};
use lance_core::Result;
use lance_linalg::distance::DistanceType;
use lance_linalg::hip::{self, check, DeviceBuffer, HipDType, HIP_CTX};

use super::kmeans::{KMeanInit, KMeans, KMeansAlgo, KMeansAlgoFloat, KMeansParams};
use super::utils::SimpleIndex;

/// Drop-in for `KMeansAlgoFloat<Float32Type>`: the E-step goes to `lance_hip_assign`; the M-step keeps the reference code
/// (it is bandwidth-trivial and defines the f32 accumulation order the engine itself reproduces when it trains end to end).
pub struct KMeansAlgoHip;

impl KMeansAlgo<f32> for KMeansAlgoHip {
    fn compute_membership_and_dist(
        centroids: &[f32],
        data: &[f32],
        dimension: usize,
        distance_type: DistanceType,
        balance_factor: f32,
        cluster_sizes: Option<&[usize]>,
        _index: Option<&SimpleIndex>, // the HNSW shortcut (utils.rs:47-108) is not needed: the exact argmin is cheaper on the device
    ) -> (Vec<Option<u32>>, Vec<Option<f32>>) {
        // argmin_value_float_with_bias (kernels.rs:92-111): value + balance_factor * cluster_size, un-biased value returned
        let bias: Option<Vec<f32>> = cluster_sizes.map(|s| s.iter().map(|&c| balance_factor * c as f32).collect());
        hip::assign_f32(data, dimension, centroids, distance_type, bias.as_deref())
            .unwrap_or_else(|e| panic!("lance_hip_assign: {e}")) // the CPU path panics on unsupported metrics too (kmeans.rs:360)
    }

    fn to_kmeans(
        data: &[f32],
        dimension: usize,
        k: usize,
        membership: &[Option<u32>],
        cluster_sizes: &mut [usize],
        distance_type: DistanceType,
        loss: f64,
    ) -> KMeans {
        KMeansAlgoFloat::<Float32Type>::to_kmeans(data, dimension, k, membership, cluster_sizes, distance_type, loss)
    }
}

/// The initial centroids a caller handed to `KMeansParams::new(Some(centroids), ..)` (kmeans.rs:106-126 stores them as
/// `KMeanInit::Incremental(Arc<FixedSizeListArray>)`, kmeans.rs:53-58), as the flat f32 slice the device call takes.  `None` for
/// `KMeanInit::Random` (the engine then draws `kmeans_random_init`'s rows itself) and for non-f32 centroids (the caller falls back).
/// Defined here, on the reference's public fields only: the reference has no such accessor.
fn init_centroids_f32(params: &KMeansParams) -> Option<&[f32]> {
    match &params.init {
        KMeanInit::Incremental(fsl) => fsl.values().as_primitive_opt::<Float32Type>().map(|a| a.values().as_ref()),
        KMeanInit::Random => None,
    }
}

/// `KMeans::with_centroids` (kmeans.rs:565-583, the only constructor from existing centroids the reference has) over a flat f32
/// vector that came back from the device.
fn kmeans_from_f32_centroids(centroids: Vec<f32>, dimension: usize, distance_type: DistanceType, loss: f64) -> KMeans {
    KMeans::with_centroids(Arc::new(Float32Array::from(centroids)), dimension, distance_type, loss)
}

/// `KMeans::new_with_params` (kmeans.rs:1008-1073) in one device call: flat Lloyd for k <= 256, hierarchical otherwise, same
/// parameters (`max_iters`, `tolerance`, `balance_factor`, `hierarchical_k`), same stopping rule.  Called from
/// `train_kmeans` (kmeans.rs:1309-1347) when the `hip` feature is on and the data is f32 / f16 / int8.
pub fn train_kmeans_hip(
    data: &[f32],
    dimension: usize,
    k: usize,
    params: &KMeansParams,
    seed: u64,
) -> Result<KMeans> {
    HIP_CTX.with(|ctx| {
        let n = data.len() / dimension;
        let x = ctx.upload(data)?;
        let cent = DeviceBuffer::<f32>::alloc(ctx, k * dimension)?;
        let (mut loss, mut iters, mut k_out) = (0f64, 0u32, 0u32);
        let init = init_centroids_f32(params);
        let init_dev = init.map(|c| ctx.upload(c)).transpose()?;
        check(unsafe {
            hip::lance_hip_kmeans_train_ex(
                ctx.as_ptr(), HipDType::F32 as i32, hip::metric_code(params.distance_type)?, x.ptr, n as u64, dimension as u32,
                k as u32, params.max_iters, params.tolerance, params.balance_factor, params.hierarchical_k as u32,
                init_dev.as_ref().map_or(std::ptr::null(), |b| b.ptr as *const _), seed, cent.ptr, &mut loss, &mut iters, &mut k_out,
            )
        })?;
        let mut centroids = cent.download()?;
        centroids.truncate(k_out as usize * dimension);
        Ok(kmeans_from_f32_centroids(centroids, dimension, params.distance_type, loss))
    })
}
