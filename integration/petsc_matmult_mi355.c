/*
 * petsc_matmult_mi355.c — the MI355X product behind PETSc's MatMult, for the reference's solver seam.
 *
 * The reference installs its hand-written SpMV into the Newton/GMRES loop with
 *     MatSetOperation(mat, MATOP_MULT, (void(*)(void))MatMult_SeqBAIJ_4_AVX2);      src/solve_newton.c:864-879
 * on MATSEQBAIJ matrices of block size 4 (S at :1061-1064, J at :1149-1152); KSPSolve (:1265) then calls it once per
 * GMRES iteration.  This file is the same seam for libmi355spmv.so: OverrideMatMultWithMI355(mat) installs
 * MatMult_MI355, which keeps one device handle per Mat (composed onto the Mat as a PetscContainer), refreshes the
 * handle's values when the Mat's object state has moved (the Newton loop rewrites J every iteration:
 * MatCopy + add_nonlinear_jacobian_terms + MatZeroRows, :1245-1247) and multiplies on the GPU.
 *
 * PETSc stores a BAIJ block COLUMN-major (src/kernels/baij4_mad.c:73-76 reads v[0], v[4], v[8], v[12] for row 0);
 * the arrays go to mi_bcsr4_create_layout(..., MI_BLOCK_COLMAJOR, ...) as they are.
 *
 * Build (only where PETSc is installed — it is NOT in this repository's image, so this file is shipped as source and
 * is compiled by nothing here; it needs the same private header the reference's kernels include,
 * src/include/kernels.h:8):
 *     mpicc -c petsc_matmult_mi355.c -I$PETSC_DIR/include -I$PETSC_DIR/$PETSC_ARCH/include -I<repo>/include
 *     ... link the solver with -L<repo>/navierstokes_amd/csrc -lmi355spmv
 * In src/solve_newton.c replace OverrideMatMultWithAVX2(J) by OverrideMatMultWithMI355(J) (same signature).
 *
 * Vectors.  Two branches per product (round 4):
 *   - DEVICE vectors (VECSEQHIP / VECHIP, a PETSc configured --with-hip; run the solver with -vec_type hip): VecHIPGetArrayRead /
 *     VecHIPGetArrayWrite hand out the HBM pointers and the product is mi_bcsr4_spmv_dev / mi_spmv_dev on them — no PCIe traffic per
 *     GMRES iteration: the kernel's own rate (bench.py: 112-120 us per product of the 1.3 M-row FE matrix, 137-150 us at C4);
 *   - HOST vectors (VECSEQ, the reference's default: VecCreateSeq, src/solve_newton.c:972): x goes in and y comes out over PCIe on
 *     every product — bench.py's `pcie_inclusive`: 1.64 ms per call at 5 M rows = 92 GFLOP/s, 11x the kernel.  Correct, and still
 *     4-5x the reference's CPU kernel, but the seam to use for speed is the device branch.
 *
 * Arithmetic: each row of y is ONE fma chain over the row's blocks in storage order, columns 0..3 inside a block —
 * the bits of mpk's SpMV_BCSR_FMA (mpk/SpMV.cpp:150-178).  MatMult_SeqBAIJ_4_AVX2 keeps four per-column accumulators
 * and adds them at the end (src/kernels/baij4_avx2.c:42-66): the two agree to rounding (rel. 1e-16 per row), not bit
 * for bit, and since that kernel cannot run without PETSc its order is parity-unpinned in this repository.
 */
#include <petscmat.h>
#include <../src/mat/impls/baij/seq/baij.h> /* Mat_SeqBAIJ: i, j, a, mbs, nbs, bs2 (as src/include/kernels.h:8) */
#include <../src/mat/impls/aij/seq/aij.h>   /* Mat_SeqAIJ: i, j, a, nz (as src/include/kernels.h:9) */

#include "mi355_spmv.h"

typedef struct {
    mi_bcsr4_t        h;
    PetscObjectState  state; /* the Mat's state the device values correspond to */
} MI355MatCtx;

static PetscErrorCode MI355MatCtxDestroy(void *p)
{
    MI355MatCtx *ctx = (MI355MatCtx *)p;
    PetscFunctionBegin;
    if (ctx) {
        (void)mi_bcsr4_destroy(ctx->h);
        PetscCall(PetscFree(ctx));
    }
    PetscFunctionReturn(PETSC_SUCCESS);
}

/* is this Vec a HIP device vector?  (type names as of PETSc 3.18+: VECSEQHIP, VECMPIHIP, VECHIP) */
static PetscErrorCode MI355VecOnDevice(Vec v, PetscBool *on)
{
    PetscFunctionBegin;
    *on = PETSC_FALSE;
#if defined(PETSC_HAVE_HIP) /* a PETSc built without HIP vectors has neither the types nor VecHIPGetArray* */
    PetscCall(PetscObjectTypeCompareAny((PetscObject)v, on, VECSEQHIP, VECMPIHIP, VECHIP, ""));
#endif
    PetscFunctionReturn(PETSC_SUCCESS);
}

#define MI_CALL(expr)                                                                                          \
    do {                                                                                                       \
        int mi_rc_ = (expr);                                                                                   \
        if (mi_rc_ != MI_OK) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "libmi355spmv: %s (%s)", mi_strerror(mi_rc_), mi_last_error()); \
    } while (0)

PetscErrorCode MatMult_MI355(Mat A, Vec xx, Vec zz)
{
    Mat_SeqBAIJ       *a = (Mat_SeqBAIJ *)A->data;
    PetscContainer     box = NULL;
    MI355MatCtx       *ctx = NULL;
    PetscObjectState   st;
    const PetscScalar *x;
    PetscScalar       *z;
    PetscBool          xdev, zdev;

    PetscFunctionBegin;
    PetscCheck(a->bs2 == 16, PETSC_COMM_SELF, PETSC_ERR_ARG_WRONG, "MatMult_MI355 needs block size 4");
    PetscCheck(sizeof(PetscInt) == sizeof(int) && sizeof(PetscScalar) == sizeof(double), PETSC_COMM_SELF, PETSC_ERR_SUP,
               "libmi355spmv takes int32 indices and real double values");
    PetscCall(PetscObjectStateGet((PetscObject)A, &st));
    PetscCall(PetscObjectQuery((PetscObject)A, "mi355_matmult_ctx", (PetscObject *)&box));
    if (!box) { /* first product with this Mat: upload pattern and values (a->i is the full row pointer even when
                   compressedrow.use is set; rows without blocks simply produce 0, as baij4_avx2.c:27-29 arranges) */
        PetscCall(PetscNew(&ctx));
        MI_CALL(mi_bcsr4_create_layout((int)a->mbs, (int)a->nbs, (const int *)a->i, (const int *)a->j, (const double *)a->a,
                                       MI_BLOCK_COLMAJOR, &ctx->h));
        ctx->state = st;
        PetscCall(PetscContainerCreate(PETSC_COMM_SELF, &box));
        PetscCall(PetscContainerSetPointer(box, ctx));
        PetscCall(PetscContainerSetUserDestroy(box, MI355MatCtxDestroy));
        PetscCall(PetscObjectCompose((PetscObject)A, "mi355_matmult_ctx", (PetscObject)box));
        PetscCall(PetscContainerDestroy(&box)); /* the Mat holds the reference now */
    } else {
        PetscCall(PetscContainerGetPointer(box, (void **)&ctx));
        if (ctx->state != st) { /* values changed under an unchanged pattern (MatCopy ... SAME_NONZERO_PATTERN) */
            MI_CALL(mi_bcsr4_update_values_layout(ctx->h, (const double *)a->a, MI_BLOCK_COLMAJOR));
            ctx->state = st;
        }
    }
    PetscCall(MI355VecOnDevice(xx, &xdev));
    PetscCall(MI355VecOnDevice(zz, &zdev));
    if (xdev && zdev) {
#if defined(PETSC_HAVE_HIP)
        /* device-resident vectors: the product runs where they live.  The launch goes to the null stream between two device
         * synchronisations (a few microseconds each; PETSc's own work runs on its PetscDeviceContext's stream — a caller that wants the
         * product ON that stream passes the hipStream_t of PetscDeviceContextGetStreamHandle as the last argument and drops both) */
        PetscCall(VecHIPGetArrayRead(xx, &x));
        PetscCall(VecHIPGetArrayWrite(zz, &z));
        MI_CALL(mi_device_synchronize());
        MI_CALL(mi_bcsr4_spmv_dev(ctx->h, (const double *)x, (double *)z, NULL));
        MI_CALL(mi_device_synchronize());
        PetscCall(VecHIPRestoreArrayRead(xx, &x));
        PetscCall(VecHIPRestoreArrayWrite(zz, &z));
#endif
    } else {
        PetscCall(VecGetArrayRead(xx, &x));
        PetscCall(VecGetArrayWrite(zz, &z));
        MI_CALL(mi_bcsr4_spmv(ctx->h, (const double *)x, (double *)z)); /* host vectors (VECSEQ): x in, y out over PCIe */
        PetscCall(VecRestoreArrayRead(xx, &x));
        PetscCall(VecRestoreArrayWrite(zz, &z));
    }
    PetscCall(PetscLogFlops(2.0 * a->nz * a->bs2 - 4.0 * a->nonzerorowcnt)); /* as src/kernels/baij4_avx2.c:82 */
    PetscFunctionReturn(PETSC_SUCCESS);
}

/* same shape as OverrideMatMultWithAVX2, src/solve_newton.c:864-879 */
PetscErrorCode OverrideMatMultWithMI355(Mat mat)
{
    PetscBool is_seqbaij;

    PetscFunctionBegin;
    PetscCall(PetscObjectTypeCompare((PetscObject)mat, MATSEQBAIJ, &is_seqbaij));
    PetscCheck(is_seqbaij, PETSC_COMM_WORLD, PETSC_ERR_ARG_WRONG, "Matrix must be of type MATSEQBAIJ");
    PetscCall(MatSetOperation(mat, MATOP_MULT, (void (*)(void))MatMult_MI355));
    PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- MATSEQAIJ: the seam of src/kernels/aij_mad.c:8-32 / aij_fma.c (MatMult_SeqAIJ, MatMult_SeqAIJ_FMA; variants 0 and 1 of
 * src/main.c:113-123 and src/kernels/variant_selector.c:3-15).  The CSR arrays go to mi_csr_create as they are; each row of y is
 * the sequential fma chain in storage order — bit-equal to MatMult_SeqAIJ_FMA's `sum = fma(aa[j], x[aj[j]], sum)` loop wherever
 * that kernel really emits fma (its mpk twin SpMV_CSR_FMA is what the oracle is pinned to), within 1e-15 of the mul-then-add of
 * aij_mad.c. */
typedef struct {
    mi_csr_t          h;
    PetscObjectState  state;
} MI355AijCtx;

static PetscErrorCode MI355AijCtxDestroy(void *p)
{
    MI355AijCtx *ctx = (MI355AijCtx *)p;
    PetscFunctionBegin;
    if (ctx) {
        (void)mi_csr_destroy(ctx->h);
        PetscCall(PetscFree(ctx));
    }
    PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode MatMult_MI355_AIJ(Mat A, Vec xx, Vec zz)
{
    Mat_SeqAIJ        *a = (Mat_SeqAIJ *)A->data;
    PetscContainer     box = NULL;
    MI355AijCtx       *ctx = NULL;
    PetscObjectState   st;
    const PetscScalar *x;
    PetscScalar       *z;
    PetscBool          xdev, zdev;

    PetscFunctionBegin;
    PetscCheck(sizeof(PetscInt) == sizeof(int) && sizeof(PetscScalar) == sizeof(double), PETSC_COMM_SELF, PETSC_ERR_SUP,
               "libmi355spmv takes int32 indices and real double values");
    PetscCall(PetscObjectStateGet((PetscObject)A, &st));
    PetscCall(PetscObjectQuery((PetscObject)A, "mi355_matmult_aij_ctx", (PetscObject *)&box));
    if (!box) {
        PetscCall(PetscNew(&ctx));
        MI_CALL(mi_csr_create((int)A->rmap->n, (int)A->cmap->n, (const int *)a->i, (const int *)a->j, (const double *)a->a, &ctx->h));
        ctx->state = st;
        PetscCall(PetscContainerCreate(PETSC_COMM_SELF, &box));
        PetscCall(PetscContainerSetPointer(box, ctx));
        PetscCall(PetscContainerSetUserDestroy(box, MI355AijCtxDestroy));
        PetscCall(PetscObjectCompose((PetscObject)A, "mi355_matmult_aij_ctx", (PetscObject)box));
        PetscCall(PetscContainerDestroy(&box));
    } else {
        PetscCall(PetscContainerGetPointer(box, (void **)&ctx));
        if (ctx->state != st) {
            MI_CALL(mi_csr_update_values(ctx->h, (const double *)a->a));
            ctx->state = st;
        }
    }
    PetscCall(MI355VecOnDevice(xx, &xdev));
    PetscCall(MI355VecOnDevice(zz, &zdev));
    if (xdev && zdev) {
#if defined(PETSC_HAVE_HIP)
        PetscCall(VecHIPGetArrayRead(xx, &x));
        PetscCall(VecHIPGetArrayWrite(zz, &z));
        MI_CALL(mi_device_synchronize());
        MI_CALL(mi_spmv_dev(ctx->h, (const double *)x, (double *)z, NULL)); /* see MatMult_MI355 for the stream */
        MI_CALL(mi_device_synchronize());
        PetscCall(VecHIPRestoreArrayRead(xx, &x));
        PetscCall(VecHIPRestoreArrayWrite(zz, &z));
#endif
    } else {
        PetscCall(VecGetArrayRead(xx, &x));
        PetscCall(VecGetArrayWrite(zz, &z));
        MI_CALL(mi_spmv(ctx->h, (const double *)x, (double *)z));
        PetscCall(VecRestoreArrayRead(xx, &x));
        PetscCall(VecRestoreArrayWrite(zz, &z));
    }
    PetscCall(PetscLogFlops(2.0 * a->nz)); /* as src/kernels/aij_mad.c:30 */
    PetscFunctionReturn(PETSC_SUCCESS);
}

/* One selector for both matrix types, in the place of MatMult_SeqBAIJ_4_VariantSelector (src/kernels/variant_selector.c:3-15): the
 * CPU variants 0..5 differ in how the CPU is driven; on the GPU they are one kernel per storage format. */
PetscErrorCode MatMult_MI355_Selector(Mat A, Vec xx, Vec zz)
{
    PetscBool is_baij, is_aij;

    PetscFunctionBegin;
    PetscCall(PetscObjectTypeCompare((PetscObject)A, MATSEQBAIJ, &is_baij));
    PetscCall(PetscObjectTypeCompare((PetscObject)A, MATSEQAIJ, &is_aij));
    if (is_baij && ((Mat_SeqBAIJ *)A->data)->bs2 == 16) PetscCall(MatMult_MI355(A, xx, zz));
    else if (is_aij) PetscCall(MatMult_MI355_AIJ(A, xx, zz));
    else SETERRQ(PETSC_COMM_SELF, PETSC_ERR_SUP, "MatMult_MI355_Selector: MATSEQAIJ or MATSEQBAIJ with block size 4 (convert an 8x8 BAIJ matrix with MatConvert(A, MATSEQAIJ, ...), as src/benchmark_spmv.c:185 does)");
    PetscFunctionReturn(PETSC_SUCCESS);
}
