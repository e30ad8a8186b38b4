//! Replacement body for warp_image_perspective of /root/reference/homographier/src/homographier/mod.rs:271-300
//! (cv::warpPerspective(src, M, size, INTER_LINEAR, BORDER_CONSTANT, Scalar(1,1,1,1))): same generic signature. The GPU path serves 8-bit
//! and 32-bit float elements with 1, 3 or 4 channels (CV_8UC1/3/4 - CV_8UC4 = Cmat<Vec4b> is what raster_to_mat returns and the
//! workspace warps - and CV_32FC1/3/4); any other element type is refused with the code OpenCV uses for an unsupported format.
//! NOT compiled in the build container (no Rust toolchain there).
use super::{Cmat, MatError};
use opencv::core::{DataType, Mat, Size2i, CV_32F, CV_8U};
use opencv::prelude::*;
use std::ffi::CStr;

/// mod.rs:271-300
pub fn warp_image_perspective<T: DataType>(src: &Cmat<T>, m: &Cmat<f64>, size: Option<Size2i>) -> Result<Cmat<T>, MatError> {
    let src_size = src.mat.size().map_err(|_err| MatError::Unknown)?; // mod.rs:277
    let size = size.unwrap_or(src_size); // mod.rs:276
    let (depth, channels) = (src.mat.depth(), src.mat.channels());
    if !(depth == CV_8U || depth == CV_32F) || !(channels == 1 || channels == 3 || channels == 4) || !src.mat.is_continuous() {
        return Err(MatError::Opencv(opencv::Error::new(-210, "apds warp_image_perspective: source must be a continuous 8U or 32F matrix of 1, 3 or 4 channels")));
    }
    if m.mat.rows() != 3 || m.mat.cols() != 3 || !m.mat.is_continuous() {
        return Err(MatError::Opencv(opencv::Error::new(-215, "the transformation matrix must be 3x3")));
    }
    // the reference allocates the destination at the SOURCE size filled with (1,1,1,1) and lets warpPerspective re-create it at `size`
    // (mod.rs:278-296): the result is a size.height x size.width image
    let mut dst = unsafe { Mat::new_rows_cols(size.height, size.width, src.mat.typ()) }.map_err(MatError::Opencv)?;
    let rc = unsafe {
        if depth == CV_8U {
            apds_sys::apds_warp_perspective(src.mat.data(), src_size.height, src_size.width, channels, m.mat.data() as *const f64, size.height, size.width,
                                            dst.data_mut())
        } else {
            apds_sys::apds_warp_perspective_f32(src.mat.data() as *const f32, src_size.height, src_size.width, channels, m.mat.data() as *const f64,
                                                size.height, size.width, dst.data_mut() as *mut f32)
        }
    };
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(apds_sys::apds_last_error()) }.to_string_lossy().into_owned();
        return Err(MatError::Opencv(opencv::Error::new(rc, msg)));
    }
    Cmat::<T>::new(dst)
}
