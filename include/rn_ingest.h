/*
 * rn_ingest.h -- C ABI of the trajectory reader that feeds the PotGNN evaluator (SURVEY.md 8f
 * item 4: "trajectory ingest").  Host-only; same shared library (librn_potgnn.so).
 *
 * Replaces the line-by-line Python parser of VASP XDATCAR trajectories,
 *   read_positions_ts            ramannoodle/io/vasp/xdatcar.py:21-56
 *   _read_lattice                ramannoodle/io/vasp/poscar.py:18-43   (comment, scale, 3 vectors)
 *   _read_atomic_symbols         ramannoodle/io/vasp/poscar.py:46-79   (symbols, counts)
 *   _read_positions              ramannoodle/io/vasp/poscar.py:82-121  (label line + N rows)
 * with the same acceptance rules: a frame is a label line whose first character is d/D
 * (direct), c/C (cartesian) or s/S (selective dynamics: the next line is the label) followed by
 * N rows of which the first three whitespace-separated tokens are parsed as Python `float`
 * does; reading ends silently at the first label line that is empty or starts with whitespace
 * (that is how the reference detects the end of the file); anything else is an invalid file.
 * Frames are indexed when the file is opened and parsed on demand, in parallel, straight into
 * the caller's (possibly pinned) buffer, so that parsing chunk k+1 overlaps evaluating chunk k.
 */
#ifndef RN_INGEST_H
#define RN_INGEST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rn_xdatcar rn_xdatcar; /* opaque */

enum {
  RN_INGEST_OK = 0,
  RN_INGEST_INVALID_ARGUMENT = -1,
  RN_INGEST_FILE_NOT_FOUND = -6, /* -> FileNotFoundError */
  RN_INGEST_INVALID_FILE = -7    /* -> InvalidFileException; text from rn_xdatcar_last_error */
};

/* Opens `path`, parses the header and indexes the frames.  On RN_INGEST_INVALID_FILE *out is
 * still a valid handle whose only use is rn_xdatcar_last_error / rn_xdatcar_close. */
int rn_xdatcar_open(const char *path, rn_xdatcar **out);
void rn_xdatcar_close(rn_xdatcar *h);

/* lattice: row-major 3x3 in Angstrom, already multiplied by the scale factor. */
int rn_xdatcar_info(const rn_xdatcar *h, int64_t *num_frames, int32_t *num_atoms, double *lattice,
                    int32_t *num_species);
/* symbol: at least 8 bytes, NUL-terminated on return. */
int rn_xdatcar_species(const rn_xdatcar *h, int32_t index, char *symbol, int32_t *count);

/*
 * Parses frames [first, first + count) into positions[count][N][3] exactly as written in the file
 * (no wrapping).  cartesian[k] (may be NULL) = 1 where frame first+k was given in Cartesian
 * coordinates: the caller converts those with positions @ inv(lattice) as the reference does.
 * num_threads <= 0: one thread per 64 frames, at most 16.
 */
int rn_xdatcar_read(rn_xdatcar *h, int64_t first, int64_t count, double *positions, uint8_t *cartesian,
                    int num_threads);

/* Message of the last failure on `h`, in the reference's wording (valid until the next call). */
const char *rn_xdatcar_last_error(const rn_xdatcar *h);

#ifdef __cplusplus
}
#endif
#endif /* RN_INGEST_H */
