//! Batched form of `feature_extraction_to_database` of /root/reference/preprocessor/src/main.rs:249-327 for the tile loop at
//! main.rs:227-245: instead of one rayon task per tile doing to_rgb -> raster_to_mat -> akaze_keypoint_descriptor_extraction_def
//! (main.rs:258-277: three crate calls, three PCIe round trips, one launch-bound small extraction), one task takes a GROUP of tiles,
//! reads their band windows (the same `rasterband.read_as::<f32>` calls `to_rgb` makes,
//! geotiff_extractor/src/image_extractor/mod.rs:318-343) and hands them to `apds_tile_extract_batch`: ONE band_merger pass and ONE
//! batched extraction for the group; the 8-bit image exists on the GPU only. Per-tile keypoints and descriptors are exactly those of
//! the three separate calls. The database rows are then written per tile as before (main.rs:279-324).
//! NOT compiled in the build container (no Rust toolchain there).
use feature_extraction::DbKeypoints;
use std::ffi::CStr;
use std::ptr;

/// One tile's band windows, row-major `rows x cols` f32, as `to_rgb` reads them for an equal-size window.
pub struct TileBands {
    pub red: Vec<f32>,
    pub green: Vec<f32>,
    pub blue: Vec<f32>,
}

/// Selects the GPU the calling rayon thread uses (one thread group per GPU). Call once per thread before the first extraction.
pub fn bind_thread_to_gpu(ordinal: i32) -> Result<(), String> {
    let rc = unsafe { apds_sys::apds_set_device(ordinal) };
    if rc != 0 {
        return Err(unsafe { CStr::from_ptr(apds_sys::apds_last_error()) }.to_string_lossy().into_owned());
    }
    Ok(())
}

/// Keypoints of a group of equal-sized tiles, as `to_db_type(image_id)` would return them per tile (lib.rs:34-58), before the
/// level-of-detail lift of main.rs:296-304. `min_max` = [red_min, red_max, green_min, green_max, blue_min, blue_max] of the dataset
/// (`datasets_min_max`). `image_ids[i]` is the id `create_image` returned for tile i.
pub fn extract_tiles(tiles: &[TileBands], rows: i32, cols: i32, min_max: &[f64; 6], image_ids: &[i32]) -> Result<Vec<Vec<DbKeypoints>>, String> {
    assert_eq!(tiles.len(), image_ids.len());
    let n = tiles.len();
    if n == 0 {
        return Ok(Vec::new());
    }
    let px = (rows as usize) * (cols as usize);
    for t in tiles {
        assert!(t.red.len() == px && t.green.len() == px && t.blue.len() == px, "band windows must be rows x cols");
    }
    let red: Vec<*const f32> = tiles.iter().map(|t| t.red.as_ptr()).collect();
    let green: Vec<*const f32> = tiles.iter().map(|t| t.green.as_ptr()).collect();
    let blue: Vec<*const f32> = tiles.iter().map(|t| t.blue.as_ptr()).collect();
    let mut counts = vec![0i32; n];
    let (mut kps, mut desc, mut nb) = (ptr::null_mut(), ptr::null_mut(), 0i32);
    let rc = unsafe {
        apds_sys::apds_tile_extract_batch(red.as_ptr(), green.as_ptr(), blue.as_ptr(), n as i32, rows, cols, cols as usize, min_max.as_ptr(), 0, &mut kps,
                                          &mut desc, counts.as_mut_ptr(), &mut nb)
    };
    if rc != 0 {
        return Err(unsafe { CStr::from_ptr(apds_sys::apds_last_error()) }.to_string_lossy().into_owned());
    }
    let total: usize = counts.iter().map(|&c| c as usize).sum();
    let k = unsafe { std::slice::from_raw_parts(kps, total) };
    let d = unsafe { std::slice::from_raw_parts(desc, total * nb as usize) };
    let mut out = Vec::with_capacity(n);
    let mut off = 0usize;
    for (i, &c) in counts.iter().enumerate() {
        let mut v = Vec::with_capacity(c as usize);
        for j in off..off + c as usize {
            v.push(DbKeypoints {
                x_coord: k[j].x,
                y_coord: k[j].y,
                size: k[j].size,
                angle: k[j].angle,
                response: k[j].response,
                octave: k[j].octave,
                class_id: k[j].class_id,
                descriptor: d[j * nb as usize..(j + 1) * nb as usize].to_vec(),
                image_id: image_ids[i],
            });
        }
        off += c as usize;
        out.push(v);
    }
    unsafe {
        apds_sys::apds_free(kps as *mut _);
        apds_sys::apds_free(desc as *mut _);
    }
    Ok(out)
}
