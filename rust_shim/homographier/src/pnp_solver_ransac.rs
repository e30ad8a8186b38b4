//! Replacement body for pnp_solver_ransac of /root/reference/homographier/src/homographier/mod.rs:320-369.
//! NOT compiled in the build container (no Rust toolchain there).
use super::{Cmat, ImgObjCorrespondence, MatError, PNPRANSACSolution};
use opencv::calib3d::SolvePnPMethod;
use opencv::core::Mat;
use opencv::prelude::*;
use std::ffi::CStr;

/// mod.rs:320-369. `dist_coeffs` never reached OpenCV in the reference either (mod.rs:344 shadows it with zeros(4,1)).
pub fn pnp_solver_ransac(point_correspondences: &[ImgObjCorrespondence], camera_intrinsic: &Cmat<f64>, iter_count: i32, reproj_thres: f32,
                         confidence: f64, _dist_coeffs: Option<&[f64]>, method: Option<SolvePnPMethod>)
    -> Result<Option<PNPRANSACSolution>, MatError> {
    // Point3d / Point2d are #[repr(C)] {f64; 3} / {f64; 2}: unzip into two flat arrays as the reference does (mod.rs:329-335)
    let n = point_correspondences.len();
    let mut obj = Vec::with_capacity(3 * n);
    let mut img = Vec::with_capacity(2 * n);
    for p in point_correspondences {
        obj.extend_from_slice(&[p.obj_point.x, p.obj_point.y, p.obj_point.z]);
        img.extend_from_slice(&[p.img_point.x, p.img_point.y]);
    }
    let k = camera_intrinsic.mat.data_typed::<f64>().map_err(MatError::Opencv)?; // 3x3 row major
    let (mut rvec, mut tvec) = ([0f64; 3], [0f64; 3]);
    let mut inliers = vec![0i32; n.max(1)];
    let (mut n_inliers, mut found) = (0i32, 0i32);
    let rc = unsafe {
        apds_sys::apds_pnp_solver_ransac(obj.as_ptr(), img.as_ptr(), n as i32, k.as_ptr(), iter_count, reproj_thres, confidence,
                                         method.unwrap_or(SolvePnPMethod::SOLVEPNP_EPNP) as i32, // mod.rs:360
                                         rvec.as_mut_ptr(), tvec.as_mut_ptr(), inliers.as_mut_ptr(), &mut n_inliers, &mut found)
    };
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(apds_sys::apds_last_error()) }.to_string_lossy().into_owned();
        return Err(MatError::Opencv(opencv::Error::new(rc, msg)));
    }
    if found == 0 {
        return Ok(None); // res.then_some(solution), mod.rs:367
    }
    inliers.truncate(n_inliers as usize);
    Ok(Some(PNPRANSACSolution {
        rvec: Cmat::new(Mat::from_slice_rows_cols(&rvec, 3, 1).map_err(MatError::Opencv)?)?,
        tvec: Cmat::new(Mat::from_slice_rows_cols(&tvec, 3, 1).map_err(MatError::Opencv)?)?,
        inliers: Cmat::new(Mat::from_slice_rows_cols(&inliers, inliers.len(), 1).map_err(MatError::Opencv)?)?,
    }))
}
