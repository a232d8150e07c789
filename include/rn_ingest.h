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
  RN_INGEST_INVALID_FILE = -7,   /* -> InvalidFileException; text from rn_*_last_error          */
  RN_INGEST_VALUE_ERROR = -8     /* -> ValueError (vasprun.xml: float() of a token failed, ragged
                                    rows: the reference lets Python's ValueError through)       */
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

/* ------------------------------------------------------------------ vasprun.xml
 * Replaces the ElementTree walk of ramannoodle/io/vasp/vasprun.py for the trajectory path:
 *   read_trajectory        vasprun.py:298-330  (un-named <structure> children of the root: the
 *                                               first <varray> child's rows; POTIM time step)
 *   _parse_positions       vasprun.py:53-74    ([float(i) for i in child.text.split()] per row)
 *   _parse_timestep        vasprun.py:278-295  (./parameters/separator[@name='ionic']/i[@name='POTIM'])
 *   read_positions / read_ref_structure   vasprun.py:217-275 (structure[@name='initialpos'];
 *                                               _parse_lattice :94-115, _parse_atomic_symbols :33-50)
 * The document is tokenised once when it is opened (ill-formed XML -> RN_INGEST_INVALID_FILE with
 * the reference's "root xml element could not be found"); frames are parsed on demand, in
 * parallel, into the caller's buffer, like rn_xdatcar_read.
 */
typedef struct rn_vasprun rn_vasprun; /* opaque */

/* On RN_INGEST_INVALID_FILE *out is still a valid handle for rn_vasprun_last_error / _close. */
int rn_vasprun_open(const char *path, rn_vasprun **out);
void rn_vasprun_close(rn_vasprun *h);

/* num_frames: un-named <structure> children of the root; num_atoms: rows of the first one (-1 if it
 * has no <varray>; reading it then fails with the reference's "structure varray not found"). */
int rn_vasprun_info(const rn_vasprun *h, int64_t *num_frames, int32_t *num_atoms);

/* Frames [first, first + count) -> positions[count][num_atoms][3], as written in the file. */
int rn_vasprun_read(rn_vasprun *h, int64_t first, int64_t count, double *positions, int num_threads);

/* POTIM in fs ("timestep not found" / "potim element has no text" as the reference). */
int rn_vasprun_timestep(rn_vasprun *h, double *timestep);

/* Rows of structure[@name='initialpos']/varray, or -1 when there is none. */
int64_t rn_vasprun_initial_num_atoms(rn_vasprun *h);

/*
 * The initial structure; every output may be NULL (skipped).  lattice: row-major 3x3 Angstrom;
 * positions: [num_atoms][3] fractional (positions_capacity in doubles); symbols: the per-atom
 * element symbols, one per line ('\n'-separated, NUL-terminated; symbols_capacity in bytes).
 * *num_atoms = number of symbols when symbols != NULL, else the rows of the positions.
 */
int rn_vasprun_initial_structure(rn_vasprun *h, int32_t *num_atoms, double *lattice, double *positions,
                                 int64_t positions_capacity, char *symbols, int64_t symbols_capacity);

const char *rn_vasprun_last_error(const rn_vasprun *h);

#ifdef __cplusplus
}
#endif
#endif /* RN_INGEST_H */
