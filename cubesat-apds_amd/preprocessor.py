"""Host-side mirror of the reference's tile scheduler (/root/reference/preprocessor/src/main.rs:175-327): every
level of detail of a mosaic is cut into tiles of one fixed pixel size, each tile goes through to_rgb -> raster_to_mat -> AKAZE,
and its keypoints are stored with their coordinates lifted back to level-0 pixels. The reference writes to Postgres through a
rayon pool; here the rows go to the GPU-resident KeypointTable (feature_database.py) and an in-memory image table."""
from . import feature_extraction, homographier


class ImageTable:
    """feature_database `ref_image` rows (models.rs InsertImage: level_of_detail, x_start, x_end, y_start, y_end); ids are 1-based
    serials as Postgres assigns them (imagedb.rs create_image returns the new id)."""

    def __init__(self):
        self.rows = []

    def create_image(self, level_of_detail, x_start, x_end, y_start, y_end):
        self.rows.append(dict(id=len(self.rows) + 1, level_of_detail=int(level_of_detail), x_start=int(x_start), x_end=int(x_end),
                              y_start=int(y_start), y_end=int(y_end)))
        return self.rows[-1]["id"]


def tile_grid(image_resolution, amount_lod, lod):
    """main.rs:212-216 — (tile_size, columns, rows) of level `lod` out of `amount_lod` levels (integer arithmetic as in the reference)."""
    tile_size = (image_resolution[0] // 2 ** (amount_lod - 1), image_resolution[1] // 2 ** (amount_lod - 1))
    columns = image_resolution[0] // (tile_size[0] * 2 ** lod)
    rows = image_resolution[1] // (tile_size[1] * 2 ** lod)
    return tile_size, columns, rows


def extract_tile(dataset, tile_size, column, row, lod, fused=True):
    """main.rs:258-277 — the part of one tile that does not touch the database: read, convert, extract. Thread-safe (the C ABI
    gives every calling thread its own stream and workspace, as the reference's rayon workers each own their OpenCV objects)."""
    span = (tile_size[0] * 2 ** lod, tile_size[1] * 2 ** lod)
    if fused:
        # the same three steps inside the library: the band windows go to the GPU once and the 8-bit image never comes back
        win = dataset.window((column * span[0], row * span[1]), span, tile_size)
        return feature_extraction.tile_keypoint_descriptor_extraction(win[0], win[1], win[2], dataset.datasets_min_max(), None)
    tile = dataset.to_rgb((column * span[0], row * span[1]), span, tile_size)                     # main.rs:258-272
    tile_mat = homographier.raster_to_mat(tile, tile_size[0], tile_size[1])                       # main.rs:274
    return feature_extraction.akaze_keypoint_descriptor_extraction_def(tile_mat.mat, None)        # main.rs:277


def store_tile(table, images, keypoints, tile_size, column, row, lod):
    """main.rs:280-324 — the image row and its keypoints. Returns (image_id, n_keypoints)."""
    span = (tile_size[0] * 2 ** lod, tile_size[1] * 2 ** lod)
    image_id = images.create_image(lod, column * span[0], column * span[0] + span[0] - 1,          # main.rs:280-293
                                   row * span[1], row * span[1] + span[1] - 1)
    # main.rs:296-324: x = x * 2^lod + column * tile_w * 2^lod (same for y), one multi-row INSERT
    table.create_keypoints(keypoints, image_id, lod, column, row, tile_size)
    return image_id, len(keypoints.keypoints)


def feature_extraction_to_database(table, images, dataset, tile_size, column, row, lod, fused=True):
    """main.rs:248-327 — one tile: read, convert, extract, insert the image row and its keypoints. Returns (image_id, n_keypoints)."""
    return store_tile(table, images, extract_tile(dataset, tile_size, column, row, lod, fused), tile_size, column, row, lod)


def downscale_from_lod(table, images, dataset, amount_lod, lod, workers=1, fused=True, batch=1):
    """main.rs:197-246 — every tile of one level. The reference spawns the tiles on a rayon pool (main.rs:233-243) and image ids
    follow whatever order the inserts reach Postgres in; here the rows are stored in row-major tile order, so ids and table contents
    depend neither on `workers` nor on `batch`. Small tiles are launch-latency-bound on the GPU, so either `batch` tiles go through
    ONE library call (apds_tile_extract_batch: every kernel's grid covers all of them; the preferred form, one host thread) or
    `workers` threads extract tiles concurrently on their own streams."""
    tile_size, columns, rows = tile_grid(dataset.raster_size(), amount_lod, lod)
    cells = [(j, i) for i in range(rows) for j in range(columns)]
    if batch > 1:
        span = (tile_size[0] * 2 ** lod, tile_size[1] * 2 ** lod)
        out = []
        for k in range(0, len(cells), batch):
            group = cells[k:k + batch]
            wins = [dataset.window((j * span[0], i * span[1]), span, tile_size) for j, i in group]
            extracted = feature_extraction.tiles_keypoint_descriptor_extraction(wins, dataset.datasets_min_max(), None)
            for (j, i), kp in zip(group, extracted):
                out.append(store_tile(table, images, kp, tile_size, j, i, lod))
        return out
    if workers <= 1:
        return [feature_extraction_to_database(table, images, dataset, tile_size, j, i, lod, fused) for j, i in cells]
    from concurrent.futures import ThreadPoolExecutor, wait
    import threading
    from ._lib import lib

    def work(cell):
        return extract_tile(dataset, tile_size, cell[0], cell[1], lod, fused)

    # every pool thread gives its stream + device workspace back before the pool goes away: the barrier makes each of the
    # `workers` threads take exactly one release task
    barrier = threading.Barrier(workers)

    def release():
        # generous, but finite: if one of the `workers` release tasks never reaches the barrier (a pool thread that could not start, an
        # exception in front of the wait) the others must not block the pool's shutdown for ever; the thread's context is released either way
        try:
            barrier.wait(timeout=120.0)
        except threading.BrokenBarrierError:
            pass
        finally:
            lib().apds_thread_release()

    out, pending = [], []
    with ThreadPoolExecutor(max_workers=workers) as pool:
        try:
            # bounded look-ahead: at most 2 x workers extracted tiles wait for their turn to be stored
            for cell in cells:
                pending.append((cell, pool.submit(work, cell)))
                if len(pending) >= 2 * workers:
                    (j, i), fut = pending.pop(0)
                    out.append(store_tile(table, images, fut.result(), tile_size, j, i, lod))
            while pending:
                (j, i), fut = pending.pop(0)
                out.append(store_tile(table, images, fut.result(), tile_size, j, i, lod))
        finally:
            # on an error above tiles may still be running: drain them first (cancel what has not started), so that every pool thread
            # is free to take its release task and the barrier cannot be left waiting; a failing release never replaces the
            # exception that brought us here
            for _, fut in pending:
                fut.cancel()
            wait([fut for _, fut in pending])
            for f in [pool.submit(release) for _ in range(workers)]:
                try:
                    f.result()
                except Exception:   # noqa: BLE001
                    pass
    return out


def process_lod_from_mosaic(table, images, dataset, lod, workers=1, fused=True, batch=1):
    """main.rs:175-194 — all levels 0 .. lod-1."""
    return [downscale_from_lod(table, images, dataset, lod, i, workers, fused, batch) for i in range(lod)]
