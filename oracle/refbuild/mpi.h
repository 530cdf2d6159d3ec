/* Test-infrastructure stub (NOT an MPI implementation).
 *
 * Single-rank stand-in for <mpi.h>, used ONLY to compile the unmodified
 * reference translation unit (/root/reference/main.cpp) into oracle/_ref/ so
 * that the oracle restatement and the golden vectors can be pinned against the
 * reference itself without depending on a system MPI.  World size is always 1:
 * in-place reductions are identity, gathers are copies, point-to-point traffic
 * cannot occur (every neighbour block is local) and aborts if attempted.
 *
 * The set of symbols is exactly what main.cpp names (grep MPI_ main.cpp).
 */
#ifndef CUP3D_ORACLE_MPI_STUB_H
#define CUP3D_ORACLE_MPI_STUB_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>

typedef int MPI_Comm;
typedef int MPI_Datatype; /* value = element size in bytes (0 = derived) */
typedef int MPI_Op;
typedef int MPI_Request;
typedef int MPI_Info;
typedef long MPI_Aint;
typedef long long MPI_Offset;
typedef struct { int MPI_SOURCE, MPI_TAG, MPI_ERROR, count; } MPI_Status;
typedef struct { FILE *f; } *MPI_File;

#define MPI_COMM_WORLD 0
#define MPI_SUCCESS 0
#define MPI_IN_PLACE ((void *)1)
#define MPI_STATUS_IGNORE ((MPI_Status *)0)
#define MPI_STATUSES_IGNORE ((MPI_Status *)0)
#define MPI_INFO_NULL 0
#define MPI_PROC_NULL (-1)
#define MPI_THREAD_FUNNELED 1
#define MPI_MODE_WRONLY 1
#define MPI_MODE_CREATE 2
#define MPI_SUM 1
#define MPI_MAX 2
#define MPI_BYTE 1
#define MPI_INT 4
#define MPI_FLOAT 4
#define MPI_LONG 8
#define MPI_LONG_LONG 8
#define MPI_DOUBLE 8
#define MPI_LONG_DOUBLE 16

#define CUP3D_MPI_STUB_DIE(name)                                               \
  do {                                                                         \
    fprintf(stderr, "oracle/_ref: MPI stub '%s' called; the single-rank stub "  \
                    "cannot do point-to-point traffic\n", name);               \
    abort();                                                                   \
  } while (0)

static inline int MPI_Init_thread(int *, char ***, int required, int *provided) {
  if (provided) *provided = required;
  return 0;
}
static inline int MPI_Finalize(void) { return 0; }
static inline int MPI_Abort(MPI_Comm, int code) { fflush(0); exit(code ? code : 1); return 0; }
static inline int MPI_Comm_size(MPI_Comm, int *s) { *s = 1; return 0; }
static inline int MPI_Comm_rank(MPI_Comm, int *r) { *r = 0; return 0; }
static inline int MPI_Barrier(MPI_Comm) { return 0; }

static inline int cup3d_stub_copy(const void *s, void *r, int count, MPI_Datatype t) {
  if (s != MPI_IN_PLACE && s != r) {
    if (t <= 0) CUP3D_MPI_STUB_DIE("copy of derived datatype");
    memcpy(r, s, (size_t)count * (size_t)t);
  }
  return 0;
}
static inline int MPI_Allreduce(const void *s, void *r, int n, MPI_Datatype t, MPI_Op, MPI_Comm) { return cup3d_stub_copy(s, r, n, t); }
#ifndef CUP3D_STUB_COUNT_IALLREDUCE
#define CUP3D_STUB_COUNT_IALLREDUCE(n) ((void)0)
#endif
static inline int MPI_Iallreduce(const void *s, void *r, int n, MPI_Datatype t, MPI_Op, MPI_Comm, MPI_Request *q) { CUP3D_STUB_COUNT_IALLREDUCE(n); *q = 0; return cup3d_stub_copy(s, r, n, t); }
static inline int MPI_Reduce(const void *s, void *r, int n, MPI_Datatype t, MPI_Op, int, MPI_Comm) { return cup3d_stub_copy(s, r, n, t); }
static inline int MPI_Allgather(const void *s, int n, MPI_Datatype t, void *r, int, MPI_Datatype, MPI_Comm) { return cup3d_stub_copy(s, r, n, t); }
static inline int MPI_Allgatherv(const void *s, int n, MPI_Datatype t, void *r, const int *, const int *, MPI_Datatype, MPI_Comm) { return cup3d_stub_copy(s, r, n, t); } /* the HIP drop-in shim */
static inline int MPI_Iallgather(const void *s, int n, MPI_Datatype t, void *r, int, MPI_Datatype, MPI_Comm, MPI_Request *q) { *q = 0; return cup3d_stub_copy(s, r, n, t); }
static inline int MPI_Bcast(void *, int, MPI_Datatype, int, MPI_Comm) { return 0; } /* used by the HIP drop-in shim only */
static inline int MPI_Exscan(const void *, void *, int, MPI_Datatype, MPI_Op, MPI_Comm) { return 0; /* rank 0: recvbuf undefined by the standard */ }

/* traffic to MPI_PROC_NULL is a no-op by the standard (LoadBalancer, main.cpp:4822-4835); messages of
 * rank 0 to itself (FluxCorrectionMPI 2898-2944, UpdateBoundary) are matched by tag through a small queue */
typedef struct cup3d_stub_msg { int tag; size_t bytes; void *data; void *recvbuf; struct cup3d_stub_msg *next; } cup3d_stub_msg;
static cup3d_stub_msg *cup3d_stub_sends = 0, *cup3d_stub_recvs = 0;
static inline int MPI_Isend(const void *buf, int n, MPI_Datatype t, int peer, int tag, MPI_Comm, MPI_Request *q) {
  *q = 0;
  if (peer == MPI_PROC_NULL) return 0;
  if (peer != 0 || t <= 0) CUP3D_MPI_STUB_DIE("MPI_Isend");
  const size_t bytes = (size_t)n * (size_t)t;
  for (cup3d_stub_msg **p = &cup3d_stub_recvs; *p; p = &(*p)->next)
    if ((*p)->tag == tag) { /* a receive is already posted */
      cup3d_stub_msg *r = *p;
      if (bytes > r->bytes) CUP3D_MPI_STUB_DIE("MPI_Isend (truncation)");
      memcpy(r->recvbuf, buf, bytes);
      *p = r->next;
      free(r);
      return 0;
    }
  cup3d_stub_msg *m = (cup3d_stub_msg *)malloc(sizeof *m);
  m->tag = tag; m->bytes = bytes; m->data = malloc(bytes ? bytes : 1); memcpy(m->data, buf, bytes); m->recvbuf = 0; m->next = 0;
  cup3d_stub_msg **p = &cup3d_stub_sends;
  while (*p) p = &(*p)->next;
  *p = m;
  return 0;
}
static inline int MPI_Irecv(void *buf, int n, MPI_Datatype t, int peer, int tag, MPI_Comm, MPI_Request *q) {
  *q = 0;
  if (peer == MPI_PROC_NULL) return 0;
  if (peer != 0 || t <= 0) CUP3D_MPI_STUB_DIE("MPI_Irecv");
  const size_t bytes = (size_t)n * (size_t)t;
  for (cup3d_stub_msg **p = &cup3d_stub_sends; *p; p = &(*p)->next)
    if ((*p)->tag == tag) { /* the matching send was posted first */
      cup3d_stub_msg *m = *p;
      if (m->bytes > bytes) CUP3D_MPI_STUB_DIE("MPI_Irecv (truncation)");
      memcpy(buf, m->data, m->bytes);
      *p = m->next;
      free(m->data);
      free(m);
      return 0;
    }
  cup3d_stub_msg *r = (cup3d_stub_msg *)malloc(sizeof *r);
  r->tag = tag; r->bytes = bytes; r->data = 0; r->recvbuf = buf; r->next = 0;
  cup3d_stub_msg **p = &cup3d_stub_recvs;
  while (*p) p = &(*p)->next;
  *p = r;
  return 0;
}
static inline int MPI_Probe(int, int, MPI_Comm, MPI_Status *) { CUP3D_MPI_STUB_DIE("MPI_Probe"); return 0; }
static inline int MPI_Get_count(const MPI_Status *, MPI_Datatype, int *) { CUP3D_MPI_STUB_DIE("MPI_Get_count"); return 0; }
static inline int MPI_Wait(MPI_Request *, MPI_Status *) { return 0; }
static inline int MPI_Waitall(int, MPI_Request *, MPI_Status *) { return 0; }
static inline int MPI_Test(MPI_Request *, int *flag, MPI_Status *) { *flag = 1; return 0; }

static inline int MPI_Type_create_struct(int, const int *, const MPI_Aint *, const MPI_Datatype *, MPI_Datatype *nt) { *nt = 0; return 0; }
static inline int MPI_Type_commit(MPI_Datatype *) { return 0; }
static inline int MPI_Type_free(MPI_Datatype *) { return 0; }

static inline int MPI_File_open(MPI_Comm, const char *path, int, MPI_Info, MPI_File *fh) {
  *fh = (MPI_File)malloc(sizeof(**fh));
  (*fh)->f = fopen(path, "wb");
  return (*fh)->f ? 0 : 1;
}
static inline int MPI_File_write_at_all(MPI_File fh, MPI_Offset off, const void *buf, int n, MPI_Datatype t, MPI_Status *) {
  fseek(fh->f, (long)off, SEEK_SET);
  fwrite(buf, (size_t)t, (size_t)n, fh->f);
  return 0;
}
static inline int MPI_File_close(MPI_File *fh) { fclose((*fh)->f); free(*fh); *fh = 0; return 0; }
#endif
