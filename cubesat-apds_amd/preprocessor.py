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


def feature_extraction_to_database(table, images, dataset, tile_size, column, row, lod):
    """main.rs:248-327 — one tile: read, convert, extract, insert the image row and its keypoints. Returns (image_id, n_keypoints)."""
    span = (tile_size[0] * 2 ** lod, tile_size[1] * 2 ** lod)
    tile = dataset.to_rgb((column * span[0], row * span[1]), span, tile_size)                     # main.rs:258-272
    tile_mat = homographier.raster_to_mat(tile, tile_size[0], tile_size[1])                       # main.rs:274
    keypoints = feature_extraction.akaze_keypoint_descriptor_extraction_def(tile_mat.mat, None)   # main.rs:277
    image_id = images.create_image(lod, column * span[0], column * span[0] + span[0] - 1,          # main.rs:280-293
                                   row * span[1], row * span[1] + span[1] - 1)
    # main.rs:296-324: x = x * 2^lod + column * tile_w * 2^lod (same for y), one multi-row INSERT
    table.create_keypoints(keypoints, image_id, lod, column, row, tile_size)
    return image_id, len(keypoints.keypoints)


def downscale_from_lod(table, images, dataset, amount_lod, lod):
    """main.rs:197-246 — every tile of one level, row-major (the reference spawns them on a thread pool: order is not part of
    the result, image ids are)."""
    tile_size, columns, rows = tile_grid(dataset.raster_size(), amount_lod, lod)
    out = []
    for i in range(rows):
        for j in range(columns):
            out.append(feature_extraction_to_database(table, images, dataset, tile_size, j, i, lod))
    return out


def process_lod_from_mosaic(table, images, dataset, lod):
    """main.rs:175-194 — all levels 0 .. lod-1."""
    return [downscale_from_lod(table, images, dataset, lod, i) for i in range(lod)]
