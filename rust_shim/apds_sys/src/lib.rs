//! Raw bindings of include/apds.h (hand-written: the header is 25 functions of plain pointers and sizes).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_float, c_int, c_void};

pub const APDS_ERR_EMPTY: c_int = -1000;
pub const APDS_ERR_BAD_ARG: c_int = -5;

/// cv::KeyPoint layout (28 bytes)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct apds_keypoint {
    pub x: f32,
    pub y: f32,
    pub size: f32,
    pub angle: f32,
    pub response: f32,
    pub octave: i32,
    pub class_id: i32,
}

/// cv::DMatch layout (16 bytes)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct apds_dmatch {
    pub query_idx: i32,
    pub train_idx: i32,
    pub img_idx: i32,
    pub distance: f32,
}

extern "C" {
    pub fn apds_last_error() -> *const c_char;
    pub fn apds_free(p: *mut c_void);
    pub fn apds_akaze_extract(img: *const u8, rows: c_int, cols: c_int, channels: c_int, stride_bytes: usize, max_points: c_int,
                              kps: *mut *mut apds_keypoint, desc: *mut *mut u8, n: *mut c_int, desc_bytes: *mut c_int) -> c_int;
    pub fn apds_get_knn_matches(origin: *const u8, n_origin: c_int, target: *const u8, n_target: c_int, desc_bytes: c_int, k: c_int,
                                filter_strength: c_float, matches: *mut *mut apds_dmatch, n_matches: *mut c_int) -> c_int;
    pub fn apds_get_bruteforce_matches(origin: *const u8, n_origin: c_int, target: *const u8, n_target: c_int, desc_bytes: c_int,
                                       matches: *mut *mut apds_dmatch, n_matches: *mut c_int) -> c_int;
    pub fn apds_get_points_from_matches(kp1: *const apds_keypoint, n1: c_int, kp2: *const apds_keypoint, n2: c_int,
                                        matches: *const apds_dmatch, n_matches: c_int, bug_compatible: c_int, pts1: *mut f32, pts2: *mut f32) -> c_int;
    pub fn apds_find_homography(input_xy: *const f32, reference_xy: *const f32, n: c_int, method: c_int, reproj_threshold: c_double,
                                h: *mut f64, mask: *mut u8) -> c_int;
    pub fn apds_raster_to_mat(rgba: *const u8, n_pixels: usize, w: c_int, h: c_int, bgra: *mut u8) -> c_int;
    /// to_rgb (equal-size window) + raster_to_mat + AKAZE of one preprocessor tile in one call (include/apds.h).
    pub fn apds_tile_extract(red: *const f32, green: *const f32, blue: *const f32, rows: c_int, cols: c_int, row_stride: usize,
                             minmax6: *const f64, max_points: c_int, kps: *mut *mut apds_keypoint, desc: *mut *mut u8, n: *mut c_int,
                             desc_bytes: *mut c_int) -> c_int;
    pub fn apds_get_world_coordinates(xy: *const f64, n: c_int, dataset_gt: *const f64, elevation_gt: *const f64, elevation: *const f64,
                                      ew: c_int, eh: c_int, xyz: *mut f64) -> c_int;
    pub fn apds_thread_release() -> c_int;
    pub fn apds_pnp_solver_ransac(obj_xyz: *const f64, img_xy: *const f64, n: c_int, camera_intrinsic: *const f64, iter_count: c_int,
                                  reproj_thres: f32, confidence: c_double, method: c_int, rvec: *mut f64, tvec: *mut f64,
                                  inliers: *mut i32, n_inliers: *mut c_int, found: *mut c_int) -> c_int;
}
