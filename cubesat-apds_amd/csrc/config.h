// csrc/config.h — every run-time switch of the library, read ONCE per process from the environment (first use) into one struct.
// The defaults are the measured best; every switch keeps results bit-identical (the parity tests run the forced variants:
// tests/test_strip_kernels_gpu.py). Round 3 folded 38 scattered getenv sites into this table (19 switches) and deleted the variants that had lost every
// measurement: the persistent-grid match kernel, the staged keypoint pipeline, the 32- and 128-pixel Hessian tiles, and the tuning knobs
// of the match's work-item plan (now constants in match_hamming.hip).
#pragma once

namespace apds {

struct Config {
    // ---- AKAZE extraction: which kernel family serves a level (1 = by level size (default), 0 = never, 2 = every level: tests)
    int nld_strip;        // APDS_NLD_STRIP    FED steps on register strips (unfused path)
    int sf_strip;         // APDS_SF_STRIP     smoothing + conductivity on register strips (unfused path)
    int base_strip;       // APDS_BASE_STRIP   image -> gray -> Lt[0] + gradient magnitude in one pass
    int level_strip;      // APDS_LEVEL_STRIP  smoothing + conductivity + first FED steps of a level on register strips (levels >= 1 Mpx)
    int level_fuse;       // APDS_LEVEL_FUSE   one launch per level through LDS (levels <= 1 Mpx)
    int level_stream;     // APDS_LEVEL_STREAM the same level step as a streaming kernel (levels >= 8 Mpx); APDS_LEVEL_STREAM_ROWS: its band height (test hook)
    int level_stream_rows;
    int doh_strip;        // APDS_DOH_STRIP    streaming Hessian / extrema kernel (levels >= 8 Mpx)
    int doh_strip_rows;   // APDS_DOH_STRIP_ROWS  band height of that kernel (0 = chosen by level size); test hook
    int kp_ranked;        // APDS_KP_RANKED    1: candidates place themselves (default); 0: two passes over the masks
    int kp_xcd;           // APDS_KP_XCD       1: every XCD takes one contiguous eighth of the keypoints in the orientation / descriptor kernels (default); 0: blocks stride over all of them
    int early_fork;       // APDS_EARLY_FORK   1: level 0's Hessian kernel may start as soon as the base pass is done, beside the contrast-factor pass; 0 (default): after it
    int half_fuse;        // APDS_HALF_FUSE    1: the launch that finishes an octave's last level also writes the next octave's start image (default); 0: half_sample_kernel
    int fed_shrink;       // APDS_FED_SHRINK   1: level_fused_kernel's FED steps skip the patches outside the zone the tile still depends on (default); 0: every step sweeps the whole region
    // ---- AKAZE extraction: scheduling
    int akaze_fork;       // APDS_AKAZE_FORK   Hessian kernels on a side stream: 1 when the caller is the only library thread, 0 never, 2 always
    int side_probe;       // APDS_SIDE_PROBE   1: pick the side stream by a one-time concurrency probe; 0: the first stream created
    int match_mfma;       // APDS_MATCH_MFMA   1: Hamming top-1 / top-2 on the FP4 matrix pipe (hamming_mfma.hip, default); 0: the vector-ALU kernel
    int early_count;       // APDS_EARLY_COUNT 1: an extraction call returns when its keypoint count is known (orientation and descriptors still running on
                           // the stream; the thread's next call on another stream waits for them); 0: when the stream is idle
    int l2_prio;           // APDS_L2_PRIO 1: the bf16 screen's waves raise their priority for a block's MFMAs and drop it for the epilogue; 0: priority 3 throughout
    int match_mfma_lds_pad; // APDS_MATCH_MFMA_LDS_PAD bytes of unused dynamic LDS per matrix-core match workgroup (occupancy experiments)
    int match_mfma_prio;   // APDS_MATCH_MFMA_PRIO wave priority (s_setprio 0..3, default 2) for the twelve MFMAs of a block; back to 0 for the ranking
    int match_mfma_sample; // APDS_MATCH_MFMA_SAMPLE rows of the matrix-core matcher's threshold launch (at most a sixteenth of the set; 0: none)
    int match_mfma_splits; // APDS_MATCH_MFMA_SPLITS n > 0: that many train-row splits instead of the fill model's count (experiments)
    int match_mfma_xcd;   // APDS_MATCH_MFMA_XCD 1 (default): a multiple of eight train-row splits pinned to the XCDs when the fill model puts it within 5 % of its best count; 0: fill model alone
    int flag_fork;        // APDS_FLAG_FORK    1: the Hessian stream waits for a value the main chain's next kernel stores; 0 (default): fork events
    int event_scope;      // APDS_EVENT_SCOPE  2: fork / join events without the system-scope fence (default); 1: the runtime's default event
    int debug_host_time;  // APDS_DEBUG_HOST_TIME  N > 0: print the host's enqueue time per extraction call every N calls (stderr)
    // ---- Hamming match
    int match_lds_cap;    // APDS_MATCH_LDS_CAP  initial occupancy cap of the main scan (bytes of unused LDS per workgroup; apds_dev_match_lds_cap)
    int match_sample;     // APDS_MATCH_SAMPLE   rows of the threshold pre-pass (16384; 0 = no pre-pass)
    // ---- homography / PnP / L2
    int ransac_batch;     // APDS_RANSAC_BATCH first speculated batch of RANSAC hypotheses (512)
    int pnp_batch;        // APDS_PNP_BATCH    hypotheses per PnP batch (2048)
    int l2_sample_div;    // APDS_L2_SAMPLE_DIV  the bf16 screen's threshold sample = rows / this (12)
    // ---- streamed frame pipeline (apds_pipeline_*)
    int pipe_extract_workers;   // APDS_EXTRACT_WORKERS  extraction threads when the caller's params say 0 (2)
    int pipe_match_split;       // APDS_MATCH_SPLIT      1: pre-pass / main scan / merge of consecutive frames on three streams (one GPU)
    int pipe_adaptive_cap;      // APDS_ADAPTIVE_CAP     1: the starvation watch may cap the main scan's occupancy
    int pipe_prio;              // APDS_PIPE_PRIO        stream priority of the short-kernel stages (-1 = high)
    // ---- test hooks
    int loopback_lag_rank, loopback_lag_ms;   // APDS_TEST_LOOPBACK_LAG="rank:ms"  that rank sleeps before the closing waits of every loopback collective
};

const Config& config();

}  // namespace apds
