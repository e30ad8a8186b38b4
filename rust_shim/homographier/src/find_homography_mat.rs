//! Replacement bodies for two functions of /root/reference/homographier/src/homographier/mod.rs; everything else in
//! that module (Cmat, MatError, HomographyMethod, warp_image_perspective) stays as it is; pnp_solver_ransac.rs replaces a third.
//! NOT compiled in the build container (no Rust toolchain there).
use super::{Cmat, HomographyMethod, MatError};
use opencv::core::{Mat, Point2f, Vec4b};
use opencv::prelude::*;
use rgb::RGBA8;
use std::ffi::CStr;

fn apds_error(code: i32) -> MatError {
    if code == apds_sys::APDS_ERR_EMPTY {
        return MatError::Empty; // empty model -> Cmat::new(empty Mat) failed in the reference (mod.rs:258,114-119)
    }
    let msg = unsafe { CStr::from_ptr(apds_sys::apds_last_error()) }.to_string_lossy().into_owned();
    MatError::Opencv(opencv::Error::new(code, msg))
}

/// mod.rs:231-259
pub fn find_homography_mat(input: &[Point2f], reference: &[Point2f], method: Option<HomographyMethod>, reproj_threshold: Option<f64>)
    -> Result<(Cmat<f64>, Option<Cmat<u8>>), MatError> {
    if input.len() != reference.len() {
        return Err(MatError::Opencv(opencv::Error::new(-215, "point lists differ in length")));
    }
    let method_i = method.unwrap_or(HomographyMethod::Default) as i32; // mod.rs:241
    let mut h = [0f64; 9];
    let mut mask = vec![0u8; input.len()];
    // Point2f is #[repr(C)] {x: f32, y: f32}: the slices are already n x 2 float arrays
    let rc = unsafe {
        apds_sys::apds_find_homography(input.as_ptr() as *const f32, reference.as_ptr() as *const f32, input.len() as i32, method_i,
                                       reproj_threshold.unwrap_or(3f64), h.as_mut_ptr(), mask.as_mut_ptr())
    };
    if rc != 0 {
        return Err(apds_error(rc));
    }
    let hm = Mat::from_slice_rows_cols(&h, 3, 3).map_err(MatError::Opencv)?;
    let out_mask = match method {
        Some(HomographyMethod::RANSAC) | Some(HomographyMethod::LMEDS) => {
            Some(Cmat::new(Mat::from_slice_rows_cols(&mask, mask.len(), 1).map_err(MatError::Opencv)?)?) // mod.rs:253-257
        }
        _ => None,
    };
    Ok((Cmat::<f64>::new(hm)?, out_mask))
}

/// mod.rs:183-197 (one kernel instead of the per-row recursion of raster_1d_to_2d, mod.rs:199-216)
pub fn raster_to_mat(pixels: &[RGBA8], w: i32, h: i32) -> Result<Cmat<Vec4b>, MatError> {
    if w <= 0 || h <= 0 || pixels.len() != (w as usize) * (h as usize) {
        return Err(MatError::Unknown); // mod.rs:185-187
    }
    let mut out = vec![Vec4b::default(); pixels.len()];
    let rc = unsafe { apds_sys::apds_raster_to_mat(pixels.as_ptr() as *const u8, pixels.len(), w, h, out.as_mut_ptr() as *mut u8) };
    if rc != 0 {
        return Err(if rc == apds_sys::APDS_ERR_BAD_ARG { MatError::Unknown } else { apds_error(rc) });
    }
    Cmat::new(Mat::from_slice_rows_cols(&out, h as usize, w as usize).map_err(MatError::Opencv)?)
}
