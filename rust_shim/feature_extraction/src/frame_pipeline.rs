//! The streamed frame pipeline of libapds_hip (`apds_pipeline_*`, include/apds.h) for a Rust host: frame -> AKAZE -> Hamming top-2 against
//! a train set resident on the GPU -> ratio test -> matched points -> homography, software-pipelined over a stream of frames by host
//! threads that live inside the library (two extraction workers | the match on three streams | ratio filter + RANSAC). The reference
//! chains `akaze_keypoint_descriptor_extraction_def` -> `get_knn_matches` -> `get_points_from_matches` only inside unit tests
//! (feature_extraction/src/lib.rs:197-249) and calls extraction from a rayon pool without a lock (preprocessor/src/main.rs:227-245); this
//! is that chain for a stream of frames, one `submit` and one `poll` per frame. NOT compiled in the build container (no Rust toolchain).
use apds_sys::{apds_frame_result, apds_pipeline_counters, apds_pipeline_params, APDS_PIPELINE_NOT_READY};
use std::ffi::CStr;
use std::os::raw::c_void;
use std::ptr;

fn err(rc: i32) -> String {
    format!("apds error {}: {}", rc, unsafe { CStr::from_ptr(apds_sys::apds_last_error()) }.to_string_lossy())
}

pub struct FrameResult {
    pub frame: i64,
    pub keypoints: i32,
    pub matches: i32,
    pub inliers: i32,
    /// row major, `h[8] == 1`; `None`: fewer than four matches survived the ratio test, or no model (`MatError::Empty`, mod.rs:258)
    pub homography: Option<[f64; 9]>,
}

pub struct FramePipeline {
    pipe: *mut c_void,
    stride: usize,
}

// the handle may move between threads; one thread submits at a time (the library serialises submitters anyway)
unsafe impl Send for FramePipeline {}

impl FramePipeline {
    /// `train_rows_dev`: n x 64-byte descriptor rows on the device (apds_dev_pack_descriptors), `train_kps_dev`: their n `KeyPoint`s (28 bytes
    /// each); both stay borrowed until the pipeline is dropped. `filter_strength`: Lowe ratio of get_knn_matches (lib.rs:107-111).
    pub fn new(gpu: i32, rows: i32, cols: i32, channels: i32, train_rows_dev: *const c_void, train_kps_dev: *const c_void, n: i64, filter_strength: f32,
               method: i32, reproj_threshold: f64) -> Result<Self, String> {
        let params = apds_pipeline_params {
            rows, cols, channels, max_points: 0, n_slots: 0, extract_workers: 0, filter_strength, homography_method: method, reproj_threshold,
            max_iters: 0, confidence: 0.0, timing: 0, match_lds_cap: 0, match_stream: ptr::null_mut(), debug_extract_delay_ms: 0.0,
        };
        let mut pipe = ptr::null_mut();
        unsafe {
            let rc = apds_sys::apds_set_device(gpu);
            if rc != 0 {
                return Err(err(rc));
            }
            let rc = apds_sys::apds_pipeline_create(&mut pipe, train_rows_dev, n, 0, ptr::null_mut(), train_kps_dev, n, &params);
            if rc != 0 {
                return Err(err(rc));
            }
        }
        Ok(FramePipeline { pipe, stride: (cols * channels) as usize })
    }

    /// Hands a frame in host memory over (the `Mat::data()` of a CV_8UC4 frame); returns its number. Blocks only while every slot is in
    /// flight. The pixels must stay valid until the frame's result has been polled.
    pub fn submit(&self, pixels: &[u8]) -> Result<i64, String> {
        let mut id = -1i64;
        let rc = unsafe { apds_sys::apds_pipeline_submit(self.pipe, pixels.as_ptr() as *const c_void, self.stride, 0, &mut id) };
        if rc != 0 {
            return Err(err(rc));
        }
        Ok(id)
    }

    /// The next result in submission order; `Ok(None)` when it is not finished yet (`wait == false`) or nothing is in flight.
    pub fn poll(&self, wait: bool) -> Result<Option<FrameResult>, String> {
        let mut r = apds_frame_result { frame: 0, status: 0, n_keypoints: 0, n_matches: 0, n_inliers: 0, homography_found: 0, H: [0.0; 9] };
        let rc = unsafe { apds_sys::apds_pipeline_poll(self.pipe, &mut r, wait as i32) };
        if rc == APDS_PIPELINE_NOT_READY {
            return Ok(None);
        }
        if rc != 0 {
            return Err(err(rc));
        }
        if r.status != 0 {
            return Err(err(r.status));
        }
        Ok(Some(FrameResult {
            frame: r.frame,
            keypoints: r.n_keypoints,
            matches: r.n_matches,
            inliers: r.n_inliers,
            homography: if r.homography_found != 0 { Some(r.H) } else { None },
        }))
    }

    pub fn frames_done(&self) -> Result<i64, String> {
        let mut c: apds_pipeline_counters = unsafe { std::mem::zeroed() };
        let rc = unsafe { apds_sys::apds_pipeline_stats(self.pipe, &mut c, 0) };
        if rc != 0 {
            return Err(err(rc));
        }
        Ok(c.frames_done)
    }
}

impl Drop for FramePipeline {
    fn drop(&mut self) {
        unsafe { apds_sys::apds_pipeline_destroy(self.pipe) };
    }
}
