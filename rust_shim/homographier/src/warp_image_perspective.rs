//! Replacement body for warp_image_perspective of /root/reference/homographier/src/homographier/mod.rs:271-300
//! (cv::warpPerspective(src, M, size, INTER_LINEAR, BORDER_CONSTANT, Scalar(1,1,1,1))): same generic signature; the GPU path serves the
//! 4-channel 8-bit case the workspace uses (Cmat<Vec4b>, what raster_to_mat returns), any other element type is refused with the
//! code OpenCV uses for an unsupported format. NOT compiled in the build container (no Rust toolchain there).
use super::{Cmat, MatError};
use opencv::core::{DataType, Mat, Size2i, CV_8UC4};
use opencv::prelude::*;
use std::ffi::CStr;

/// mod.rs:271-300
pub fn warp_image_perspective<T: DataType>(src: &Cmat<T>, m: &Cmat<f64>, size: Option<Size2i>) -> Result<Cmat<T>, MatError> {
    let src_size = src.mat.size().map_err(|_err| MatError::Unknown)?; // mod.rs:277
    let size = size.unwrap_or(src_size); // mod.rs:276
    if src.mat.typ() != CV_8UC4 || !src.mat.is_continuous() {
        return Err(MatError::Opencv(opencv::Error::new(-210, "apds warp_image_perspective: source must be a continuous CV_8UC4 matrix")));
    }
    if m.mat.rows() != 3 || m.mat.cols() != 3 || !m.mat.is_continuous() {
        return Err(MatError::Opencv(opencv::Error::new(-215, "the transformation matrix must be 3x3")));
    }
    // the reference allocates the destination at the SOURCE size filled with (1,1,1,1) and lets warpPerspective re-create it at `size`
    // (mod.rs:278-296): the result is a size.height x size.width image
    let mut dst = unsafe { Mat::new_rows_cols(size.height, size.width, src.mat.typ()) }.map_err(MatError::Opencv)?;
    let rc = unsafe {
        apds_sys::apds_warp_perspective(src.mat.data(), src_size.height, src_size.width, 4, m.mat.data() as *const f64, size.height, size.width,
                                        dst.data_mut())
    };
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(apds_sys::apds_last_error()) }.to_string_lossy().into_owned();
        return Err(MatError::Opencv(opencv::Error::new(rc, msg)));
    }
    Cmat::<T>::new(dst)
}
