/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the scan -> bitmap rasteriser.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's library.
 *
 * Restates weap_util/weap_util/lidar.py:4-103 (`_lidar_to_bitmap`; the same body is repeated in
 * src/SAL.py:274-345 and src/bitmap.py:4-94 with other grey levels).  The drawing itself lives in a
 * third-party dependency that is NOT under /root/reference and is not importable in this image:
 *   opencv-python-headless == 4.11.0.86   (pinned at weap_util/setup.py:8)
 * i.e. OpenCV 4.11.0, modules/imgproc/src/drawing.cpp.  The functions below restate that file's
 * published algorithms for the one case the reference uses (8-bit single channel, LINE_8, shift 0,
 * thickness 1 or filled):
 *   clip_line           <- cv::clipLine(Size2l, Point2l&, Point2l&)
 *   line8               <- Line() + LineIterator::init/operator++ (connectivity 8, leftToRight = true)
 *   collect_poly_edges  <- CollectPolyEdges()  (outline + edge records, incl. the clipped-endpoint correction)
 *   fill_edge_collection<- FillEdgeCollection() (sorted edge list, active list, per-scanline bubble sort)
 *   rect_filled         <- cv::rectangle(thickness < 0) -> FillConvexPoly() of an axis-aligned box
 *
 * PARITY UNPINNED: cv2 cannot be imported here, the reference holds no bitmap together with the scan
 * it was drawn from (its lidar_datasets/*.npz are point-occupancy grids of another routine), so nothing
 * in this file has been checked against OpenCV's own output.  The HIP kernel is held bit-exact to THIS
 * restatement; DESIGN.md says the same.
 *
 * The sequential structure of OpenCV (linked active-edge list, one Bresenham iterator per segment) is
 * kept on purpose: the HIP kernel uses closed forms and bit-plane parity counting instead, and the tests
 * compare the two formulations. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { XY_SHIFT = 16, XY_ONE = 1 << XY_SHIFT };

typedef struct { int64_t x, y; } P2;
typedef struct { uint8_t *data; int rows, cols; } Img;

/* drawing.cpp clipLine(Size2l, Point2l&, Point2l&) */
static int clip_line(int64_t width, int64_t height, P2 *pt1, P2 *pt2)
{
    int c1, c2;
    const int64_t right = width - 1, bottom = height - 1;
    if (width <= 0 || height <= 0) return 0;
    int64_t *x1 = &pt1->x, *y1 = &pt1->y, *x2 = &pt2->x, *y2 = &pt2->y;
    c1 = (*x1 < 0) + (*x1 > right) * 2 + (*y1 < 0) * 4 + (*y1 > bottom) * 8;
    c2 = (*x2 < 0) + (*x2 > right) * 2 + (*y2 < 0) * 4 + (*y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        int64_t a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            *x1 += (int64_t)((double)(a - *y1) * (double)(*x2 - *x1) / (double)(*y2 - *y1));
            *y1 = a;
            c1 = (*x1 < 0) + (*x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            *x2 += (int64_t)((double)(a - *y2) * (double)(*x2 - *x1) / (double)(*y2 - *y1));
            *y2 = a;
            c2 = (*x2 < 0) + (*x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                *y1 += (int64_t)((double)(a - *x1) * (double)(*y2 - *y1) / (double)(*x2 - *x1));
                *x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                *y2 += (int64_t)((double)(a - *x2) * (double)(*y2 - *y1) / (double)(*x2 - *x1));
                *x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

/* Line(): LineIterator(img, pt1, pt2, 8, leftToRight = true), every visited pixel set to color */
static void line8(Img *img, P2 p1, P2 p2, uint8_t color)
{
    P2 pt1 = p1, pt2 = p2;
    if ((uint64_t)pt1.x >= (uint64_t)img->cols || (uint64_t)pt2.x >= (uint64_t)img->cols ||
        (uint64_t)pt1.y >= (uint64_t)img->rows || (uint64_t)pt2.y >= (uint64_t)img->rows) {
        if (!clip_line(img->cols, img->rows, &pt1, &pt2)) return;
    }
    int delta_x = 1, delta_y = 1;
    int dx = (int)(pt2.x - pt1.x), dy = (int)(pt2.y - pt1.y);
    if (dx < 0) { dx = -dx; dy = -dy; pt1 = pt2; }          /* leftToRight */
    if (dy < 0) { dy = -dy; delta_y = -1; }
    const int vert = dy > dx;
    if (vert) { int t = dx; dx = dy; dy = t; t = delta_x; delta_x = delta_y; delta_y = t; }
    int err = dx - (dy + dy);
    const int plusDelta = dx + dx, minusDelta = -(dy + dy);
    int minusShift = delta_x, plusShift = 0, minusStep = 0, plusStep = delta_y;
    const int count = dx + 1;
    if (vert) { int t = plusStep; plusStep = plusShift; plusShift = t; t = minusStep; minusStep = minusShift; minusShift = t; }
    int64_t x = pt1.x, y = pt1.y;
    for (int i = 0; i < count; i++) {
        img->data[(size_t)y * img->cols + x] = color;
        const int mask = err < 0 ? -1 : 0;
        err += minusDelta + (plusDelta & mask);
        y += minusStep + (plusStep & mask);
        x += minusShift + (plusShift & mask);
    }
}

typedef struct PolyEdge { int y0, y1; int64_t x, dx; struct PolyEdge *next; } PolyEdge;

/* CollectPolyEdges(img, v, count, edges, color, LINE_8, shift = 0, offset = (0,0)); returns #edges appended */
static int collect_poly_edges(Img *img, const P2 *v, int count, PolyEdge *edges, uint8_t color)
{
    int n = 0;
    P2 pt0 = v[count - 1], pt1;
    pt0.x = pt0.x << XY_SHIFT;
    for (int i = 0; i < count; i++, pt0 = pt1) {
        P2 t0, t1;
        PolyEdge edge;
        pt1 = v[i];
        pt1.x = pt1.x << XY_SHIFT;
        P2 pt0c = pt0, pt1c = pt1;
        t0.y = pt0.y; t1.y = pt1.y;
        t0.x = (pt0.x + (XY_ONE >> 1)) >> XY_SHIFT;
        t1.x = (pt1.x + (XY_ONE >> 1)) >> XY_SHIFT;
        line8(img, t0, t1, color);
        /* use clipped endpoints to create a more accurate PolyEdge */
        if ((uint32_t)t0.x >= (uint32_t)img->cols || (uint32_t)t1.x >= (uint32_t)img->cols ||
            (uint32_t)t0.y >= (uint32_t)img->rows || (uint32_t)t1.y >= (uint32_t)img->rows) {
            clip_line(img->cols, img->rows, &t0, &t1);
            if (t0.y != t1.y) {
                pt0c.y = t0.y; pt1c.y = t1.y;
                pt0c.x = t0.x << XY_SHIFT;
                pt1c.x = t1.x << XY_SHIFT;
            }
        } else {
            pt0c.x += XY_ONE >> 1;
            pt1c.x += XY_ONE >> 1;
        }
        if (pt0.y == pt1.y) continue;
        edge.dx = (pt1c.x - pt0c.x) / (pt1c.y - pt0c.y);
        if (pt0.y < pt1.y) {
            edge.y0 = (int)pt0.y; edge.y1 = (int)pt1.y;
            edge.x = pt0c.x + (pt0.y - pt0c.y) * edge.dx;   /* correct starting point for clipped lines */
        } else {
            edge.y0 = (int)pt1.y; edge.y1 = (int)pt0.y;
            edge.x = pt1c.x + (pt1.y - pt1c.y) * edge.dx;
        }
        edge.next = 0;
        edges[n++] = edge;
    }
    return n;
}

static int cmp_edges(const void *a, const void *b)
{
    const PolyEdge *e1 = (const PolyEdge *)a, *e2 = (const PolyEdge *)b;
    if (e1->y0 != e2->y0) return e1->y0 < e2->y0 ? -1 : 1;
    if (e1->x != e2->x) return e1->x < e2->x ? -1 : 1;
    if (e1->dx != e2->dx) return e1->dx < e2->dx ? -1 : 1;
    return 0;
}

static void hline(Img *img, int y, int x1, int x2, uint8_t color)
{
    for (int x = x1; x <= x2; x++) img->data[(size_t)y * img->cols + x] = color;
}

/* FillEdgeCollection(img, edges, color, LINE_8); edges must have room for total + 1 records */
static void fill_edge_collection(Img *img, PolyEdge *edges, int total, uint8_t color)
{
    PolyEdge tmp;
    int i, y;
    PolyEdge *e;
    int y_max = INT32_MIN, y_min = INT32_MAX;
    int64_t x_max = -1, x_min = INT64_MAX;
    const int delta = 0;                                    /* line_type < LINE_AA */
    if (total < 2) return;
    for (i = 0; i < total; i++) {
        PolyEdge *e1 = &edges[i];
        const int64_t x1 = e1->x + (int64_t)(e1->y1 - e1->y0) * e1->dx;
        if (e1->y0 < y_min) y_min = e1->y0;
        if (e1->y1 > y_max) y_max = e1->y1;
        if (e1->x < x_min) x_min = e1->x;
        if (e1->x > x_max) x_max = e1->x;
        if (x1 < x_min) x_min = x1;
        if (x1 > x_max) x_max = x1;
    }
    if (y_max < 0 || y_min >= img->rows || x_max < 0 || x_min >= ((int64_t)img->cols << XY_SHIFT)) return;
    qsort(edges, (size_t)total, sizeof(PolyEdge), cmp_edges);
    memset(&tmp, 0, sizeof(tmp));
    tmp.y0 = INT32_MAX;
    edges[total] = tmp;
    i = 0;
    tmp.next = 0;
    e = &edges[i];
    if (y_max > img->rows) y_max = img->rows;
    for (y = e->y0; y < y_max; y++) {
        PolyEdge *last, *prelast, *keep_prelast;
        int draw = 0;
        const int clipline = y < 0;
        prelast = &tmp;
        last = tmp.next;
        while (last || e->y0 == y) {
            if (last && last->y1 == y) {                    /* exclude edge if y reaches its lower point */
                prelast->next = last->next;
                last = last->next;
                continue;
            }
            keep_prelast = prelast;
            if (last && (e->y0 > y || last->x < e->x)) {    /* go to the next edge in active list */
                prelast = last;
                last = last->next;
            } else if (i < total) {                         /* insert new edge into active list */
                prelast->next = e;
                e->next = last;
                prelast = e;
                e = &edges[++i];
            } else
                break;
            if (draw) {
                if (!clipline) {
                    int x1, x2;
                    if (keep_prelast->x > prelast->x) {
                        x1 = (int)((prelast->x + delta) >> XY_SHIFT);
                        x2 = (int)(keep_prelast->x >> XY_SHIFT);
                    } else {
                        x1 = (int)((keep_prelast->x + delta) >> XY_SHIFT);
                        x2 = (int)(prelast->x >> XY_SHIFT);
                    }
                    if (x1 < img->cols && x2 >= 0) {
                        if (x1 < 0) x1 = 0;
                        if (x2 >= img->cols) x2 = img->cols - 1;
                        hline(img, y, x1, x2, color);
                    }
                }
                keep_prelast->x += keep_prelast->dx;
                prelast->x += prelast->dx;
            }
            draw ^= 1;
        }
        /* sort edges (using bubble sort) */
        keep_prelast = 0;
        do {
            prelast = &tmp;
            last = tmp.next;
            PolyEdge *last_exchange = 0;
            while (last != keep_prelast && last->next != 0) {
                PolyEdge *te = last->next;
                if (last->x > te->x) {                      /* swap edges */
                    prelast->next = te;
                    last->next = te->next;
                    te->next = last;
                    prelast = te;
                    last_exchange = prelast;
                } else {
                    prelast = last;
                    last = te;
                }
            }
            if (last_exchange == 0) break;
            keep_prelast = last_exchange;
        } while (keep_prelast != tmp.next && keep_prelast != &tmp);
    }
}

/* cv::fillPoly(img, [pts], color) */
void cv_fill_poly(uint8_t *data, int rows, int cols, const int64_t *pts, int count, int color)
{
    Img img = {data, rows, cols};
    if (count <= 0) return;
    PolyEdge *edges = (PolyEdge *)malloc(sizeof(PolyEdge) * (size_t)(count + 1));
    const int n = collect_poly_edges(&img, (const P2 *)pts, count, edges, (uint8_t)color);
    fill_edge_collection(&img, edges, n, (uint8_t)color);
    free(edges);
}

/* cv::polylines(img, [pts], isClosed, color, thickness = 1): PolyLine -> ThickLine -> Line */
void cv_polylines(uint8_t *data, int rows, int cols, const int64_t *pts, int count, int closed, int color)
{
    Img img = {data, rows, cols};
    const P2 *v = (const P2 *)pts;
    if (count <= 0) return;
    int i = closed ? 0 : 1;
    P2 p0 = v[closed ? count - 1 : 0];
    for (; i < count; i++) {
        const P2 p = v[i];
        line8(&img, p0, p, (uint8_t)color);
        p0 = p;
    }
}

/* cv::line(img, p1, p2, color, thickness = 1) */
void cv_line(uint8_t *data, int rows, int cols, int64_t x1, int64_t y1, int64_t x2, int64_t y2, int color)
{
    Img img = {data, rows, cols};
    P2 a = {x1, y1}, b = {x2, y2};
    line8(&img, a, b, (uint8_t)color);
}

/* cv::rectangle(img, p1, p2, color, thickness = -1): FillConvexPoly of the 4 corners.  For an
 * axis-aligned box with integer corners its outline pass (Line on every side) plus its span pass
 * cover exactly the inclusive box clipped to the image. */
void cv_rectangle_filled(uint8_t *data, int rows, int cols, int64_t x1, int64_t y1, int64_t x2, int64_t y2, int color)
{
    Img img = {data, rows, cols};
    if (x1 > x2) { int64_t t = x1; x1 = x2; x2 = t; }
    if (y1 > y2) { int64_t t = y1; y1 = y2; y2 = t; }
    if (x2 < 0 || y2 < 0 || x1 >= cols || y1 >= rows) return;
    if (x1 < 0) x1 = 0;
    if (y1 < 0) y1 = 0;
    if (x2 >= cols) x2 = cols - 1;
    if (y2 >= rows) y2 = rows - 1;
    for (int64_t y = y1; y <= y2; y++) hline(&img, (int)y, (int)x1, (int)x2, (uint8_t)color);
}

/* lidar.py:70-84: points[k] = rint(center + (scaling_factor * data[k]) * {cos, sin}(angles[k])).astype(int).
 * idx / cosv / sinv are the host-side numpy tables (np.linspace(..., dtype=int), np.cos, np.sin). */
void lidar_points(const double *scan, const int32_t *idx, const double *cosv, const double *sinv, int T,
                  double cx, double cy, double scale, int64_t *pts)
{
    for (int k = 0; k < T; k++) {
        const double d = scale * scan[idx[k]];
        pts[2 * k + 0] = (int32_t)(int64_t)rint(cx + d * cosv[k]);   /* the cv2 binding narrows int64 -> int32 */
        pts[2 * k + 1] = (int32_t)(int64_t)rint(cy + d * sinv[k]);
    }
}

/* _lidar_to_bitmap, lidar.py:63-103.  mode: 0 FILL, 1 POLYGON, 2 RAYS.  img: rows*cols bytes. */
void lidar_bitmap(const double *scan, const int32_t *idx, const double *cosv, const double *sinv, int T,
                  int rows, int cols, double scale, int mode, int bg, int draw, int draw_center, uint8_t *img)
{
    const int64_t cx = rows / 2, cy = cols / 2;             /* center = (dims[0]//2, dims[1]//2) used as (x, y) */
    int64_t *pts = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)T);
    memset(img, bg, (size_t)rows * cols);
    lidar_points(scan, idx, cosv, sinv, T, (double)cx, (double)cy, scale, pts);
    if (mode == 0)
        cv_fill_poly(img, rows, cols, pts, T, draw);
    else if (mode == 1)
        cv_polylines(img, rows, cols, pts, T, 1, draw);
    else
        for (int k = 0; k < T; k++) {
            cv_line(img, rows, cols, cx, cy, pts[2 * k], pts[2 * k + 1], draw);
            cv_rectangle_filled(img, rows, cols, pts[2 * k] - 2, pts[2 * k + 1] - 2, pts[2 * k] + 2, pts[2 * k + 1] + 2, draw);
        }
    if (draw_center)
        cv_rectangle_filled(img, rows, cols, cx - 2, cy - 2, cx + 2, cy + 2, mode == 0 ? bg : draw);
    free(pts);
}

void lidar_bitmap_batch(const double *scans, int64_t n, int num_beams, const int32_t *idx, const double *cosv,
                        const double *sinv, int T, int rows, int cols, double scale, int mode, int bg, int draw,
                        int draw_center, uint8_t *imgs)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < n; i++)
        lidar_bitmap(scans + i * num_beams, idx, cosv, sinv, T, rows, cols, scale, mode, bg, draw, draw_center,
                     imgs + (size_t)i * rows * cols);
}

/* f1tenth_gym/examples/lidar.py:212-244: point-occupancy grid of one scan (the routine that wrote the
 * reference's lidar_datasets/*.npz).  angles: np.linspace(-135, 135, n) * pi / 180 (host numpy table,
 * with its cos / sin).  grid: grid_size^2 bytes, zeroed here. */
void lidar_occupancy(const double *scan, const double *cosv, const double *sinv, int n, double max_range,
                     double lo, double hi, int grid_size, uint8_t *grid)
{
    memset(grid, 0, (size_t)grid_size * grid_size);
    for (int b = 0; b < n; b++) {
        const double r = scan[b];
        if (r >= max_range) continue;
        const double x = r * cosv[b], y = r * sinv[b];
        if (!(lo <= x && x <= hi && lo <= y && y <= hi)) continue;
        int i_row = (int)(((x - lo) / (hi - lo)) * (grid_size - 1));
        int i_col = (int)(((y - lo) / (hi - lo)) * (grid_size - 1));
        if (i_row < 0) i_row = 0;
        if (i_row > grid_size - 1) i_row = grid_size - 1;
        if (i_col < 0) i_col = 0;
        if (i_col > grid_size - 1) i_col = grid_size - 1;
        grid[(size_t)i_row * grid_size + i_col] = 1;
    }
}
