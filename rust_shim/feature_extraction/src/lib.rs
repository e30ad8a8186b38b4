//! Drop-in replacement of /root/reference/feature_extraction/src/lib.rs: same public names, signatures and error type;
//! every body forwards to libapds_hip.so (include/apds.h). NOT compiled in the build container (no Rust toolchain there):
//! keep it this small and obviously correct. Reference line numbers are cited per item.
use cv::{
    core::{DMatch, KeyPoint, Mat, Point2f, Vector, CV_8U},
    Error,
};
use opencv::{self as cv, prelude::*};
use std::ffi::CStr;
use std::ptr;

pub const MAX_POINTS_SHIFT: i32 = 18; // lib.rs:12
pub const MAX_POINTS: i32 = (1 << MAX_POINTS_SHIFT) - 1; // lib.rs:13

pub struct ExtractedKeyPoint {
    keypoints: Vector<KeyPoint>,
    descriptors: Mat,
}

#[derive(Debug)]
pub struct DbKeypoints {
    pub x_coord: f32,
    pub y_coord: f32,
    pub size: f32,
    pub angle: f32,
    pub response: f32,
    pub octave: i32,
    pub class_id: i32,
    pub descriptor: Vec<u8>,
    pub image_id: i32,
}

fn apds_err(code: i32) -> Error {
    let msg = unsafe { CStr::from_ptr(apds_sys::apds_last_error()) }.to_string_lossy().into_owned();
    Error::new(code, msg)
}

fn desc_rows(m: &Mat) -> Result<(*const u8, i32, i32), Error> {
    if m.rows() == 0 {
        return Ok((ptr::null(), 0, m.cols().max(0)));
    }
    if m.typ() != CV_8U || !m.is_continuous() {
        return Err(Error::new(-215, "descriptors must be a continuous CV_8U matrix"));
    }
    Ok((m.data(), m.rows(), m.cols()))
}

impl ExtractedKeyPoint {
    /// lib.rs:34-58
    pub fn to_db_type(&self, image_id: i32) -> Vec<DbKeypoints> {
        self.keypoints
            .iter()
            .enumerate()
            .map(|(i, k)| DbKeypoints {
                x_coord: k.pt().x,
                y_coord: k.pt().y,
                size: k.size(),
                angle: k.angle(),
                response: k.response(),
                octave: k.octave(),
                class_id: k.class_id(),
                descriptor: self.descriptors.at_row::<u8>(i as i32).expect("Could not find descriptor").to_vec(),
                image_id,
            })
            .collect()
    }
}

/// lib.rs:61-92
pub fn akaze_keypoint_descriptor_extraction_def(img: &Mat, max_points: Option<i32>) -> Result<ExtractedKeyPoint, Error> {
    let (mut kps, mut desc, mut n, mut nb) = (ptr::null_mut(), ptr::null_mut(), 0i32, 0i32);
    let stride = img.step1(0)? * img.elem_size1();
    let rc = unsafe {
        apds_sys::apds_akaze_extract(img.data(), img.rows(), img.cols(), img.channels(), stride, max_points.unwrap_or(MAX_POINTS),
                                     &mut kps, &mut desc, &mut n, &mut nb)
    };
    if rc != 0 {
        return Err(apds_err(rc));
    }
    let mut keypoints = Vector::<KeyPoint>::with_capacity(n as usize);
    let raw = unsafe { std::slice::from_raw_parts(kps, n as usize) };
    for k in raw {
        keypoints.push(KeyPoint::new_point(Point2f::new(k.x, k.y), k.size, k.angle, k.response, k.octave, k.class_id)?);
    }
    let descriptors = if n > 0 {
        let rows = unsafe { std::slice::from_raw_parts(desc, (n * nb) as usize) };
        Mat::from_slice_rows_cols(rows, n as usize, nb as usize)? // copies
    } else {
        Mat::default()
    };
    unsafe {
        apds_sys::apds_free(kps as *mut _);
        apds_sys::apds_free(desc as *mut _);
    }
    Ok(ExtractedKeyPoint { keypoints, descriptors })
}

fn take_matches(p: *mut apds_sys::apds_dmatch, n: i32) -> Vector<DMatch> {
    let mut v = Vector::<DMatch>::with_capacity(n as usize);
    for m in unsafe { std::slice::from_raw_parts(p, n as usize) } {
        v.push(DMatch { query_idx: m.query_idx, train_idx: m.train_idx, img_idx: m.img_idx, distance: m.distance });
    }
    unsafe { apds_sys::apds_free(p as *mut _) };
    v
}

/// lib.rs:94-114
pub fn get_knn_matches(origin_desc: &Mat, target_desc: &Mat, k: i32, filter_strength: f32) -> Result<Vector<DMatch>, Error> {
    let (q, nq, qb) = desc_rows(origin_desc)?;
    let (t, nt, tb) = desc_rows(target_desc)?;
    let (mut out, mut n) = (ptr::null_mut(), 0i32);
    let rc = unsafe { apds_sys::apds_get_knn_matches(q, nq, t, nt, if nq > 0 { qb } else { tb }, k, filter_strength, &mut out, &mut n) };
    if rc != 0 {
        return Err(apds_err(rc));
    }
    Ok(take_matches(out, n))
}

/// lib.rs:116-126
pub fn get_bruteforce_matches(origin_desc: &Mat, target_desc: &Mat) -> Result<Vector<DMatch>, Error> {
    let (q, nq, qb) = desc_rows(origin_desc)?;
    let (t, nt, tb) = desc_rows(target_desc)?;
    let (mut out, mut n) = (ptr::null_mut(), 0i32);
    let rc = unsafe { apds_sys::apds_get_bruteforce_matches(q, nq, t, nt, if nq > 0 { qb } else { tb }, &mut out, &mut n) };
    if rc != 0 {
        return Err(apds_err(rc));
    }
    Ok(take_matches(out, n))
}

/// lib.rs:161-180. `bug_compatible = 1` keeps the reference's exact output (img_idx as the img1 index, img1's points
/// returned twice); switch to 0 for the intended gather once callers expect it.
pub fn get_points_from_matches(img1_keypoints: &Vector<KeyPoint>, img2_keypoints: &Vector<KeyPoint>, matches: &Vector<DMatch>)
    -> Result<(Vector<Point2f>, Vector<Point2f>), Error> {
    let conv = |v: &Vector<KeyPoint>| -> Vec<apds_sys::apds_keypoint> {
        v.iter().map(|k| apds_sys::apds_keypoint { x: k.pt().x, y: k.pt().y, size: k.size(), angle: k.angle(), response: k.response(),
                                                    octave: k.octave(), class_id: k.class_id() }).collect()
    };
    let (k1, k2) = (conv(img1_keypoints), conv(img2_keypoints));
    let m: Vec<apds_sys::apds_dmatch> = matches.iter().map(|m| apds_sys::apds_dmatch { query_idx: m.query_idx, train_idx: m.train_idx,
                                                                                       img_idx: m.img_idx, distance: m.distance }).collect();
    let mut p1 = vec![0f32; m.len() * 2];
    let mut p2 = vec![0f32; m.len() * 2];
    let rc = unsafe {
        apds_sys::apds_get_points_from_matches(k1.as_ptr(), k1.len() as i32, k2.as_ptr(), k2.len() as i32, m.as_ptr(), m.len() as i32, 1,
                                               p1.as_mut_ptr(), p2.as_mut_ptr())
    };
    if rc != 0 {
        return Err(apds_err(rc));
    }
    let to_vec = |p: &[f32]| p.chunks(2).map(|c| Point2f::new(c[0], c[1])).collect::<Vector<Point2f>>();
    Ok((to_vec(&p1), to_vec(&p2)))
}

// export_matches / get_mat_from_dir (lib.rs:128-159) are debug image I/O and stay on OpenCV unchanged.
