//! A train set resident on the GPUs of one node, for a host that matches frame after frame against the same database rows.
//! The reference re-reads its train set from Postgres (`Keypoint::read_keypoints_from_lod` etc., feature_database/src/keypointdb.rs:50-90)
//! and hands it to `get_knn_matches` as a Mat (feature_extraction/src/lib.rs:94-114); here every rank (one host thread or process per
//! GPU) uploads ITS block of those rows once and then asks for the two nearest rows of each query over the WHOLE set
//! (`apds_shard_knn`: all-gather of the ranks' queries, local scan, all-to-all of the per-shard keys, u64-min merge = the unsharded
//! result including the lowest-index tie-break). NOT compiled in the build container (no Rust toolchain there).
use apds_sys::{apds_comm_id, APDS_TRANSPORT_LOOPBACK, APDS_TRANSPORT_RCCL};
use std::ffi::CStr;
use std::os::raw::c_void;
use std::ptr;

fn err(rc: i32) -> String {
    format!("apds error {}: {}", rc, unsafe { CStr::from_ptr(apds_sys::apds_last_error()) }.to_string_lossy())
}

pub struct ResidentShard {
    shard: *mut c_void,
    rows_dev: *mut c_void,
    pub rank: i32,
    pub world: i32,
}

/// Rank 0 calls this and passes the 128 bytes to the other ranks by any host-side means (a channel between the rank threads, a file, MPI).
pub fn new_comm_id(threads_of_one_process: bool) -> Result<apds_comm_id, String> {
    let mut id = apds_comm_id { bytes: [0; apds_sys::APDS_COMM_ID_BYTES] };
    let rc = unsafe { apds_sys::apds_comm_id_create(if threads_of_one_process { APDS_TRANSPORT_LOOPBACK } else { APDS_TRANSPORT_RCCL }, &mut id) };
    if rc != 0 {
        return Err(err(rc));
    }
    Ok(id)
}

impl ResidentShard {
    /// Collective over the `world` ranks. `descriptors`: this rank's block of the train rows, 61 bytes each (DbKeypoints::descriptor,
    /// lib.rs:29); `index_base`: the global index of its first row (trainIdx of the results is global). `gpu`: device ordinal of this rank.
    pub fn create(rank: i32, world: i32, gpu: i32, rccl: bool, id: &apds_comm_id, descriptors: &[u8], index_base: u32) -> Result<Self, String> {
        assert!(descriptors.len() % 61 == 0);
        let n = (descriptors.len() / 61) as i64;
        unsafe {
            let rc = apds_sys::apds_set_device(gpu);
            if rc != 0 {
                return Err(err(rc));
            }
            let (mut raw, mut rows) = (ptr::null_mut(), ptr::null_mut());
            for (p, bytes) in [(&mut raw, descriptors.len()), (&mut rows, n as usize * 64)] {
                let rc = apds_sys::apds_dev_alloc(bytes, p);
                if rc != 0 {
                    return Err(err(rc));
                }
            }
            let mut rc = apds_sys::apds_dev_upload(raw, descriptors.as_ptr() as *const c_void, descriptors.len(), ptr::null_mut());
            if rc == 0 {
                rc = apds_sys::apds_dev_pack_descriptors(raw, n, 61, 61, rows, ptr::null_mut()); // 61-byte rows -> one 64-byte line per row
            }
            if rc == 0 {
                rc = apds_sys::apds_stream_synchronize(ptr::null_mut());
            }
            apds_sys::apds_dev_release(raw);
            let mut shard = ptr::null_mut();
            if rc == 0 {
                rc = apds_sys::apds_shard_create(&mut shard, rank, world, if rccl { APDS_TRANSPORT_RCCL } else { APDS_TRANSPORT_LOOPBACK }, id, ptr::null(), rows, n,
                                                 index_base);
            }
            if rc != 0 {
                apds_sys::apds_dev_release(rows);
                return Err(err(rc));
            }
            Ok(ResidentShard { shard, rows_dev: rows, rank, world })
        }
    }

    /// Collective: (trainIdx, distance) of the two nearest rows of every query of THIS rank over the whole train set, in query order
    /// (what `knn_train_match_def(query, train, &mut matches, 2)` yields, lib.rs:103); `queries`: 61-byte rows.
    pub fn knn2(&self, queries: &[u8]) -> Result<Vec<[(i32, i32); 2]>, String> {
        assert!(queries.len() % 61 == 0);
        let nq = (queries.len() / 61) as i32;
        unsafe {
            let (mut raw, mut q, mut keys) = (ptr::null_mut(), ptr::null_mut(), ptr::null_mut());
            for (p, bytes) in [(&mut raw, queries.len()), (&mut q, nq as usize * 64), (&mut keys, nq as usize * 16)] {
                let rc = apds_sys::apds_dev_alloc(bytes, p);
                if rc != 0 {
                    return Err(err(rc));
                }
            }
            let mut rc = apds_sys::apds_dev_upload(raw, queries.as_ptr() as *const c_void, queries.len(), ptr::null_mut());
            if rc == 0 {
                rc = apds_sys::apds_dev_pack_descriptors(raw, nq as i64, 61, 61, q, ptr::null_mut());
            }
            if rc == 0 {
                rc = apds_sys::apds_shard_knn(self.shard, q, nq, ptr::null(), 2, keys, ptr::null_mut());
            }
            let mut host = vec![0u64; nq as usize * 2];
            if rc == 0 {
                rc = apds_sys::apds_dev_download(host.as_mut_ptr() as *mut c_void, keys, host.len() * 8, ptr::null_mut());
            }
            for p in [raw, q, keys] {
                apds_sys::apds_dev_release(p);
            }
            if rc != 0 {
                return Err(err(rc));
            }
            Ok(host.chunks(2).map(|k| [((k[0] & 0xFFFF_FFFF) as i32, (k[0] >> 32) as i32), ((k[1] & 0xFFFF_FFFF) as i32, (k[1] >> 32) as i32)]).collect())
        }
    }
}

impl Drop for ResidentShard {
    fn drop(&mut self) {
        unsafe {
            apds_sys::apds_shard_destroy(self.shard);
            apds_sys::apds_dev_release(self.rows_dev);
        }
    }
}
