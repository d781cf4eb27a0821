/*
 * dl_oracle.c -- CPU oracle of the DRLoco hot path.  TEST INFRASTRUCTURE ONLY (see dl_oracle.h).
 *
 * Plain C99, float64, one walker at a time, dense textbook formulations chosen for clarity and
 * for being *different* from the device kernels' formulations (mass matrix as sum of
 * J^T I J, bias as J^T (I Jdot v + ...), dense constraint Jacobian, dense Cholesky), so that
 * agreement between the two is evidence and not a tautology.
 *
 * Reference citations are relative to /root/reference.  [3P] marks arithmetic that lives in
 * third-party MuJoCo (not in the reference tree): restated from MuJoCo 2.x's documented pipeline
 * (computation chapter: kinematics, CRB, constraint model solref/solimp, pyramidal cones, Newton
 * solver with exact line search, RK4) for exactly the features walker3d_flat_feet.xml uses.
 * Dynamics parity is therefore UNPINNED against the real binary.
 */
#include "dl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NB DL_MAX_BODY
#define NV DL_MAX_DOF
#define MINVAL 1e-15
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define MAXVAL 1e10

/* ------------------------------------------------------------------ small linear algebra */
static void v3_set(double* r, double a, double b, double c) { r[0] = a; r[1] = b; r[2] = c; }
static void v3_copy(double* r, const double* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static void v3_add(double* r, const double* a, const double* b) { for (int i = 0; i < 3; i++) r[i] = a[i] + b[i]; }
static void v3_sub(double* r, const double* a, const double* b) { for (int i = 0; i < 3; i++) r[i] = a[i] - b[i]; }
static void v3_addscl(double* r, const double* a, const double* b, double s) { for (int i = 0; i < 3; i++) r[i] = a[i] + s * b[i]; }
static double v3_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void v3_cross(double* r, const double* a, const double* b) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
static double v3_norm(const double* a) { return sqrt(v3_dot(a, a)); }
/* r = R a (R row-major 3x3) */
static void m3_mulv(double* r, const double* R, const double* a) {
    double x = R[0] * a[0] + R[1] * a[1] + R[2] * a[2];
    double y = R[3] * a[0] + R[4] * a[1] + R[5] * a[2];
    double z = R[6] * a[0] + R[7] * a[1] + R[8] * a[2];
    r[0] = x; r[1] = y; r[2] = z;
}
static void m3_mul(double* r, const double* A, const double* B) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(r, t, sizeof t);
}
static void m3_identity(double* R) { memset(R, 0, 9 * sizeof(double)); R[0] = R[4] = R[8] = 1; }
/* rotation by angle about unit axis (Rodrigues) */
static void m3_axisangle(double* R, const double* ax, double ang) {
    double c = cos(ang), s = sin(ang), t = 1 - c, x = ax[0], y = ax[1], z = ax[2];
    R[0] = t * x * x + c;     R[1] = t * x * y - s * z; R[2] = t * x * z + s * y;
    R[3] = t * x * y + s * z; R[4] = t * y * y + c;     R[5] = t * y * z - s * x;
    R[6] = t * x * z - s * y; R[7] = t * y * z + s * x; R[8] = t * z * z + c;
}

/* dense Cholesky A = L L^T (lower), n <= NV, leading dimension NV; returns 0 on success */
static int chol_factor(int n, const double A[NV][NV], double L[NV][NV]) {
    for (int j = 0; j < n; j++) {
        double s = A[j][j];
        for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
        if (!(s > MINVAL)) return -1;
        L[j][j] = sqrt(s);
        for (int i = j + 1; i < n; i++) {
            double t = A[i][j];
            for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
            L[i][j] = t / L[j][j];
        }
    }
    return 0;
}
static void chol_solve(int n, const double L[NV][NV], const double* b, double* x) {
    double y[NV];
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i][k] * y[k];
        y[i] = s / L[i][i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = y[i];
        for (int k = i + 1; k < n; k++) s -= L[k][i] * x[k];
        x[i] = s / L[i][i];
    }
}

/* ------------------------------------------------------------------ model + per-evaluation data */
typedef struct {
    dl_model_desc m;
    unsigned char anc[NB][NV]; /* dof j moves body b */
    double xfrc[3];            /* [3P] xfrc_applied on the torso (body 1): world-frame force at its centre of mass */
} model_t;

typedef struct {
    double xpos[NB][3], xmat[NB][9], xipos[NB][3];
    double xanchor[NV][3], xaxis[NV][3];
    double M[NV][NV], LM[NV][NV];
    double bias[NV], passive[NV], actuator[NV], smooth[NV], qacc_smooth[NV], qacc[NV], qfrc_con[NV];
    double act_force[DL_MAX_ACT];
    int ncon;
    double cdist[DLO_MAXCON], cpos[DLO_MAXCON][3], cframe[DLO_MAXCON][9], cmu[DLO_MAXCON];
    int cbody[DLO_MAXCON], cgeom[DLO_MAXCON];
    int nefc;
    double J[DLO_MAXEFC][NV], epos[DLO_MAXEFC], ediag[DLO_MAXEFC], eR[DLO_MAXEFC], eD[DLO_MAXEFC];
    double eimp[DLO_MAXEFC], earef[DLO_MAXEFC], eforce[DLO_MAXEFC];
    int niter;
    double cost;
} data_t;

#define F_NOCONTACT 1
#define F_NOLIMIT 2
#define F_NODAMP 4
#define F_NOGRAV 8
#define F_NOACT 16

static void model_init(model_t* mm, const dl_model_desc* m) {
    mm->m = *m;
    memset(mm->anc, 0, sizeof mm->anc);
    memset(mm->xfrc, 0, sizeof mm->xfrc);
    for (int b = 1; b < m->nbody; b++)
        for (int j = 0; j < m->nv; j++) {
            int a = b;
            while (a > 0) {
                if (m->jnt_body[j] == a) { mm->anc[b][j] = 1; break; }
                a = m->body_parent[a];
            }
        }
}

/* [3P] mj_kinematics: body frames, joint anchors/axes in world coordinates */
static void kinematics(const model_t* mm, const double* q, data_t* d) {
    const dl_model_desc* m = &mm->m;
    v3_set(d->xpos[0], 0, 0, 0);
    m3_identity(d->xmat[0]);
    v3_set(d->xipos[0], 0, 0, 0);
    for (int b = 1; b < m->nbody; b++) {
        int p = m->body_parent[b];
        double pos[3], R[9], t[3];
        m3_mulv(t, d->xmat[p], m->body_pos[b]);
        v3_add(pos, d->xpos[p], t);
        memcpy(R, d->xmat[p], sizeof R);
        for (int j = 0; j < m->nv; j++) {
            if (m->jnt_body[j] != b) continue;
            double anchor[3], axis[3];
            m3_mulv(t, R, m->jnt_pos[j]);
            v3_add(anchor, pos, t);
            m3_mulv(axis, R, m->jnt_axis[j]);
            v3_copy(d->xanchor[j], anchor);
            v3_copy(d->xaxis[j], axis);
            double dq = q[j] - m->jnt_qpos0[j];
            if (m->jnt_type[j] == DL_JNT_SLIDE) {
                v3_addscl(pos, pos, axis, dq);
            } else {
                double Rj[9];
                m3_axisangle(Rj, m->jnt_axis[j], dq);
                m3_mul(R, R, Rj);
                m3_mulv(t, R, m->jnt_pos[j]);
                v3_sub(pos, anchor, t); /* off-centre rotation correction */
            }
        }
        v3_copy(d->xpos[b], pos);
        memcpy(d->xmat[b], R, sizeof R);
        m3_mulv(t, R, m->body_ipos[b]);
        v3_add(d->xipos[b], pos, t);
    }
}

/* column j of the translational / rotational Jacobian of `point` fixed to body b */
static void jac_col(const model_t* mm, const data_t* d, const double* point, int j, double* jp, double* jr) {
    if (mm->m.jnt_type[j] == DL_JNT_SLIDE) {
        v3_copy(jp, d->xaxis[j]);
        v3_set(jr, 0, 0, 0);
    } else {
        double r[3];
        v3_sub(r, point, d->xanchor[j]);
        v3_cross(jp, d->xaxis[j], r);
        v3_copy(jr, d->xaxis[j]);
    }
}

static void jac(const model_t* mm, const data_t* d, const double* point, int b, double jp[3][NV], double jr[3][NV]) {
    for (int j = 0; j < mm->m.nv; j++) {
        double cp[3] = {0, 0, 0}, cr[3] = {0, 0, 0};
        if (mm->anc[b][j]) jac_col(mm, d, point, j, cp, cr);
        for (int k = 0; k < 3; k++) { jp[k][j] = cp[k]; jr[k][j] = cr[k]; }
    }
}

/* world-frame inertia tensor of body b about its COM */
static void world_inertia(const model_t* mm, const data_t* d, int b, double Iw[9]) {
    const double* R = d->xmat[b];
    const double* I = mm->m.body_inertia[b];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Iw[3 * i + j] = R[3 * i] * I[0] * R[3 * j] + R[3 * i + 1] * I[1] * R[3 * j + 1] + R[3 * i + 2] * I[2] * R[3 * j + 2];
}

/* [3P] mj_crb result: M = sum_b m Jp^T Jp + Jr^T Iw Jr + armature */
static void mass_matrix(const model_t* mm, data_t* d) {
    const dl_model_desc* m = &mm->m;
    int nv = m->nv;
    memset(d->M, 0, sizeof d->M);
    for (int b = 1; b < m->nbody; b++) {
        double jp[3][NV], jr[3][NV], Iw[9];
        jac(mm, d, d->xipos[b], b, jp, jr);
        world_inertia(mm, d, b, Iw);
        for (int i = 0; i < nv; i++) {
            if (!mm->anc[b][i]) continue;
            double Ijr[3] = {Iw[0] * jr[0][i] + Iw[1] * jr[1][i] + Iw[2] * jr[2][i],
                             Iw[3] * jr[0][i] + Iw[4] * jr[1][i] + Iw[5] * jr[2][i],
                             Iw[6] * jr[0][i] + Iw[7] * jr[1][i] + Iw[8] * jr[2][i]};
            for (int j = 0; j < nv; j++) {
                if (!mm->anc[b][j]) continue;
                d->M[i][j] += m->body_mass[b] * (jp[0][i] * jp[0][j] + jp[1][i] * jp[1][j] + jp[2][i] * jp[2][j]) +
                              Ijr[0] * jr[0][j] + Ijr[1] * jr[1][j] + Ijr[2] * jr[2][j];
            }
        }
    }
    for (int i = 0; i < nv; i++) d->M[i][i] += m->jnt_armature[i];
}

/* [3P] mj_rne(flg_acc=0): bias = sum_b Jp^T m (Jpdot v - g) + Jr^T (Iw Jrdot v + w x Iw w) */
static void bias_forces(const model_t* mm, const double* v, int flags, data_t* d) {
    const dl_model_desc* m = &mm->m;
    int nv = m->nv;
    double g[3] = {m->gravity[0], m->gravity[1], m->gravity[2]};
    if (flags & F_NOGRAV) v3_set(g, 0, 0, 0);
    /* per dof: angular velocity of the frame carrying axis j (before joint j), velocity of anchor j */
    double wpre[NV][3], odot[NV][3], adot[NV][3];
    for (int j = 0; j < nv; j++) {
        int b = m->jnt_body[j];
        v3_set(wpre[j], 0, 0, 0);
        v3_set(odot[j], 0, 0, 0);
        for (int k = 0; k < j; k++) {
            if (!mm->anc[b][k]) continue;
            double cp[3], cr[3];
            jac_col(mm, d, d->xanchor[j], k, cp, cr);
            v3_addscl(odot[j], odot[j], cp, v[k]);
            v3_addscl(wpre[j], wpre[j], cr, v[k]);
        }
        v3_cross(adot[j], wpre[j], d->xaxis[j]);
    }
    memset(d->bias, 0, sizeof d->bias);
    for (int b = 1; b < m->nbody; b++) {
        double jp[3][NV], jr[3][NV], Iw[9];
        const double* p = d->xipos[b];
        jac(mm, d, p, b, jp, jr);
        world_inertia(mm, d, b, Iw);
        double pdot[3] = {0, 0, 0}, w[3] = {0, 0, 0};
        for (int j = 0; j < nv; j++)
            for (int k = 0; k < 3; k++) { pdot[k] += jp[k][j] * v[j]; w[k] += jr[k][j] * v[j]; }
        double a_lin[3] = {0, 0, 0}, a_ang[3] = {0, 0, 0};
        for (int j = 0; j < nv; j++) {
            if (!mm->anc[b][j]) continue;
            if (m->jnt_type[j] == DL_JNT_SLIDE) {
                v3_addscl(a_lin, a_lin, adot[j], v[j]);
            } else {
                double r[3], rd[3], t1[3], t2[3];
                v3_sub(r, p, d->xanchor[j]);
                v3_sub(rd, pdot, odot[j]);
                v3_cross(t1, adot[j], r);
                v3_cross(t2, d->xaxis[j], rd);
                v3_addscl(a_lin, a_lin, t1, v[j]);
                v3_addscl(a_lin, a_lin, t2, v[j]);
                v3_addscl(a_ang, a_ang, adot[j], v[j]);
            }
        }
        double f[3], tq[3], Iw_w[3], Iw_a[3], wxIw[3];
        for (int k = 0; k < 3; k++) f[k] = m->body_mass[b] * (a_lin[k] - g[k]);
        m3_mulv(Iw_w, Iw, w);
        m3_mulv(Iw_a, Iw, a_ang);
        v3_cross(wxIw, w, Iw_w);
        v3_add(tq, Iw_a, wxIw);
        for (int j = 0; j < nv; j++) {
            if (!mm->anc[b][j]) continue;
            d->bias[j] += jp[0][j] * f[0] + jp[1][j] * f[1] + jp[2][j] * f[2] + jr[0][j] * tq[0] + jr[1][j] * tq[1] + jr[2][j] * tq[2];
        }
    }
}

/* [3P] mj_collision for plane-vs-{capsule,box}; floor is z = 0 with normal +z
 * (walker3d_flat_feet.xml:14; only the floor has conaffinity=1, :9) */
static void add_contact(const model_t* mm, data_t* d, int g, double dist, const double* pos, const double* yaxis_hint) {
    if (d->ncon >= DLO_MAXCON) return;
    int c = d->ncon++;
    d->cdist[c] = dist;
    v3_copy(d->cpos[c], pos);
    d->cbody[c] = mm->m.geom_body[g];
    d->cgeom[c] = g;
    double mu = mm->m.geom_friction[g] > mm->m.floor_friction ? mm->m.geom_friction[g] : mm->m.floor_friction;
    d->cmu[c] = mu;
    /* mju_makeFrame: x = normal; y = hint made orthogonal (default (0,1,0)); z = x cross y */
    double* F = d->cframe[c];
    v3_set(F, 0, 0, 1);
    double y[3] = {0, 0, 0};
    if (yaxis_hint && v3_norm(yaxis_hint) >= 0.5) v3_copy(y, yaxis_hint);
    else v3_set(y, 0, 1, 0); /* |normal.y| < 0.5 */
    double dp = v3_dot(F, y);
    v3_addscl(y, y, F, -dp);
    double n = v3_norm(y);
    if (n < MINVAL) v3_set(y, 1, 0, 0);
    else { y[0] /= n; y[1] /= n; y[2] /= n; }
    v3_copy(F + 3, y);
    v3_cross(F + 6, F, F + 3);
}

static void collide(const model_t* mm, data_t* d) {
    const dl_model_desc* m = &mm->m;
    d->ncon = 0;
    for (int g = 0; g < m->ngeom; g++) {
        int b = m->geom_body[g];
        double gpos[3], gmat[9], t[3];
        m3_mulv(t, d->xmat[b], m->geom_pos[g]);
        v3_add(gpos, d->xpos[b], t);
        m3_mul(gmat, d->xmat[b], m->geom_mat[g]);
        if (m->geom_type[g] == DL_GEOM_CAPSULE) {
            /* mjc_PlaneCapsule: two end spheres, 'to' end first; frame y along the capsule axis */
            double axis[3] = {gmat[2], gmat[5], gmat[8]};
            double rad = m->geom_size[g][0], half = m->geom_size[g][1];
            for (int s = 0; s < 2; s++) {
                double c[3];
                v3_addscl(c, gpos, axis, s == 0 ? half : -half);
                double dist = c[2] - rad;
                if (dist > 0) continue; /* margin = 0; active iff dist < margin is applied below */
                if (!(dist < 0)) continue;
                double pos[3] = {c[0], c[1], c[2] - (rad + 0.5 * dist)};
                add_contact(mm, d, g, dist, pos, axis);
            }
        } else {
            /* mjc_PlaneBox: test the 8 corners in index order, keep the first 4 below the plane */
            int cnt = 0;
            for (int i = 0; i < 8 && cnt < 4; i++) {
                double vec[3] = {(i & 1 ? 1 : -1) * m->geom_size[g][0], (i & 2 ? 1 : -1) * m->geom_size[g][1], (i & 4 ? 1 : -1) * m->geom_size[g][2]};
                double corner[3];
                m3_mulv(corner, gmat, vec);
                double ldist = corner[2];
                double dist = gpos[2] + ldist;
                if (dist > 0 || ldist > 0) continue;
                if (!(dist < 0)) continue;
                double pos[3] = {gpos[0] + corner[0], gpos[1] + corner[1], gpos[2] + corner[2] - 0.5 * dist};
                add_contact(mm, d, g, dist, pos, NULL);
                cnt++;
            }
        }
    }
}

/* [3P] solimp sigmoid (getimpedance) */
static double impedance(const double* si, double pos) {
    if (si[0] == si[1] || si[2] <= MINVAL) return 0.5 * (si[0] + si[1]);
    double x = fabs(pos / si[2]);
    if (x >= 1) return si[1];
    if (x <= 0) return si[0];
    double y;
    if (si[4] == 1) y = x;
    else if (x <= si[3]) y = pow(x, si[4]) / pow(si[3], si[4] - 1);
    else y = 1 - pow(1 - x, si[4]) / pow(1 - si[3], si[4] - 1);
    return si[0] + y * (si[1] - si[0]);
}

/* [3P] mj_makeConstraint + mj_makeImpedance + mj_referenceConstraint:
 * rows = joint limits (lower, upper) then 4 pyramid edges per contact */
static void make_constraint(const model_t* mm, const double* q, const double* v, int flags, data_t* d) {
    const dl_model_desc* m = &mm->m;
    int nv = m->nv, n = 0;
    if (!(flags & F_NOLIMIT))
        for (int j = 0; j < nv; j++) {
            if (!m->jnt_limited[j]) continue;
            for (int side = -1; side <= 1; side += 2) {
                double dist = side * (m->jnt_range[j][(side + 1) / 2] - q[j]);
                if (dist < 0) {
                    memset(d->J[n], 0, sizeof d->J[n]);
                    d->J[n][j] = -(double)side;
                    d->epos[n] = dist;
                    d->ediag[n] = m->dof_invweight0[j];
                    n++;
                }
            }
        }
    int first_contact_row = n;
    if (!(flags & F_NOCONTACT))
        for (int c = 0; c < d->ncon; c++) {
            double jp[3][NV], jr[3][NV], Jc[3][NV];
            jac(mm, d, d->cpos[c], d->cbody[c], jp, jr);
            for (int k = 0; k < 3; k++)
                for (int j = 0; j < nv; j++) Jc[k][j] = d->cframe[c][3 * k] * jp[0][j] + d->cframe[c][3 * k + 1] * jp[1][j] + d->cframe[c][3 * k + 2] * jp[2][j];
            double mu = d->cmu[c];
            double tran = m->body_invweight0[d->cbody[c]][0] + m->body_invweight0[0][0];
            for (int k = 1; k <= 2; k++)
                for (int s = 0; s < 2; s++) {
                    for (int j = 0; j < nv; j++) d->J[n][j] = Jc[0][j] + (s == 0 ? mu : -mu) * Jc[k][j];
                    d->epos[n] = d->cdist[c];
                    d->ediag[n] = tran + mu * mu * tran;
                    n++;
                }
        }
    d->nefc = n;
    /* impedance, regulariser, reference acceleration */
    double tc = m->solref[0], dr = m->solref[1], dmax = m->solimp[1];
    if (tc < 2 * m->timestep) tc = 2 * m->timestep; /* refsafe */
    double K = 1.0 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr);
    double B = 2.0 / fmax(MINVAL, dmax * tc);
    for (int i = 0; i < n; i++) {
        double imp = impedance(m->solimp, d->epos[i]);
        d->eimp[i] = imp;
        d->eR[i] = fmax(MINVAL, (1 - imp) * d->ediag[i] / imp);
    }
    if (!(flags & F_NOCONTACT))
        for (int c = 0; c < d->ncon; c++) {
            int id = first_contact_row + 4 * c;
            double Rpy = 2 * d->cmu[c] * d->cmu[c] * d->eR[id];
            for (int k = 0; k < 4; k++) d->eR[id + k] = Rpy;
        }
    for (int i = 0; i < n; i++) {
        d->eD[i] = 1.0 / d->eR[i];
        double vel = 0;
        for (int j = 0; j < nv; j++) vel += d->J[i][j] * v[j];
        d->earef[i] = -B * vel - K * d->eimp[i] * d->epos[i];
    }
}

/* ---- [3P] Newton solver on the primal problem
 *   min_a 1/2 (a - a_s)^T M (a - a_s) + sum_i 1/2 D_i min(0, (J a - aref)_i)^2            */
typedef struct { double alpha, cost, d1, d2; } lspoint;
typedef struct {
    int nv, nefc;
    const data_t* d;
    double g0, g1, g2;
    const double *Jaref, *Jv;
} lsctx;

static void ls_eval(const lsctx* c, double alpha, lspoint* p) {
    double cost = c->g0 + alpha * c->g1 + alpha * alpha * c->g2;
    double d1 = c->g1 + 2 * alpha * c->g2, d2 = 2 * c->g2;
    for (int i = 0; i < c->nefc; i++) {
        double x = c->Jaref[i] + alpha * c->Jv[i];
        if (x < 0) {
            double D = c->d->eD[i];
            cost += 0.5 * D * x * x;
            d1 += D * x * c->Jv[i];
            d2 += D * c->Jv[i] * c->Jv[i];
        }
    }
    p->alpha = alpha; p->cost = cost; p->d1 = d1; p->d2 = d2;
}

static int ls_update_bracket(const lsctx* c, lspoint* p, const lspoint cand[3], lspoint* pnext) {
    int flag = 0;
    for (int i = 0; i < 3; i++) {
        if (p->d1 < 0 && cand[i].d1 < 0 && p->d1 < cand[i].d1) { *p = cand[i]; flag = 1; }
        else if (p->d1 > 0 && cand[i].d1 > 0 && p->d1 > cand[i].d1) { *p = cand[i]; flag = 2; }
    }
    if (flag) ls_eval(c, p->alpha - p->d1 / p->d2, pnext);
    return flag;
}

/* exact line search on the piecewise quadratic (MuJoCo's scheme: Newton steps from one side until
 * the derivative changes sign, then a bracketed search over {Newton from both ends, midpoint}) */
static double linesearch(const lsctx* c, double gtol, int maxit) {
    lspoint p0, p1, p2, pmid, p1next, p2next;
    int it = 0;
    ls_eval(c, 0, &p0);
    ls_eval(c, p0.alpha - p0.d1 / p0.d2, &p1);
    if (p0.cost < p1.cost) p1 = p0;
    if (fabs(p1.d1) < gtol) return p1.alpha;
    int dir = p1.d1 < 0 ? 1 : -1, p2update = 0;
    p2 = p1;
    while (p1.d1 * dir <= -gtol && it < maxit) {
        p2 = p1;
        p2update = 1;
        ls_eval(c, p1.alpha - p1.d1 / p1.d2, &p1);
        it++;
        if (fabs(p1.d1) < gtol) return p1.alpha;
    }
    if (it >= maxit || !p2update) return p1.alpha;
    p2next = p1;
    ls_eval(c, p1.alpha - p1.d1 / p1.d2, &p1next);
    while (it < maxit) {
        ls_eval(c, 0.5 * (p1.alpha + p2.alpha), &pmid);
        it++;
        lspoint cand[3] = {p1next, p2next, pmid};
        int best = -1;
        for (int i = 0; i < 3; i++)
            if (fabs(cand[i].d1) < gtol && (best < 0 || cand[i].cost < cand[best].cost)) best = i;
        if (best >= 0) return cand[best].alpha;
        int b1 = ls_update_bracket(c, &p1, cand, &p1next);
        int b2 = ls_update_bracket(c, &p2, cand, &p2next);
        if (!b1 && !b2) return pmid.cost < p0.cost ? pmid.alpha : 0.0;
    }
    if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
    if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
    return 0.0;
}

static void mul_M(const data_t* d, int nv, const double* x, double* r) {
    for (int i = 0; i < nv; i++) {
        double s = 0;
        for (int j = 0; j < nv; j++) s += d->M[i][j] * x[j];
        r[i] = s;
    }
}
static void mul_J(const data_t* d, int nv, const double* x, double* r) {
    for (int i = 0; i < d->nefc; i++) {
        double s = 0;
        for (int j = 0; j < nv; j++) s += d->J[i][j] * x[j];
        r[i] = s;
    }
}

/* cost of a candidate acceleration (used by the warmstart choice) */
static double total_cost(const data_t* d, int nv, const double* a) {
    double Ma[NV], Ja[DLO_MAXEFC], cost = 0;
    mul_M(d, nv, a, Ma);
    mul_J(d, nv, a, Ja);
    for (int i = 0; i < d->nefc; i++) {
        double x = Ja[i] - d->earef[i];
        if (x < 0) cost += 0.5 * d->eD[i] * x * x;
    }
    for (int i = 0; i < nv; i++) cost += 0.5 * (Ma[i] - d->smooth[i]) * (a[i] - d->qacc_smooth[i]);
    return cost;
}

static void solver_update(data_t* d, int nv, const double* Ma, const double* Jaref, double* gauss, double H[NV][NV], double L[NV][NV], double* grad, double* Mgrad) {
    double cost = 0;
    memset(d->qfrc_con, 0, sizeof d->qfrc_con);
    memcpy(H, d->M, sizeof d->M);
    for (int i = 0; i < d->nefc; i++) {
        if (Jaref[i] < 0) {
            double D = d->eD[i];
            d->eforce[i] = -D * Jaref[i];
            cost += 0.5 * D * Jaref[i] * Jaref[i];
            for (int j = 0; j < nv; j++) {
                d->qfrc_con[j] += d->J[i][j] * d->eforce[i];
                for (int k = 0; k < nv; k++) H[j][k] += D * d->J[i][j] * d->J[i][k];
            }
        } else d->eforce[i] = 0;
    }
    double g = 0;
    for (int i = 0; i < nv; i++) g += 0.5 * (Ma[i] - d->smooth[i]) * (d->qacc[i] - d->qacc_smooth[i]);
    *gauss = g;
    d->cost = cost + g;
    for (int i = 0; i < nv; i++) grad[i] = Ma[i] - d->smooth[i] - d->qfrc_con[i];
    chol_factor(nv, H, L);
    chol_solve(nv, L, grad, Mgrad);
}

static void solve(const model_t* mm, const double* warm, data_t* d) {
    const dl_model_desc* m = &mm->m;
    int nv = m->nv, nefc = d->nefc;
    d->niter = 0;
    if (nefc == 0) {
        memcpy(d->qacc, d->qacc_smooth, sizeof d->qacc);
        memset(d->qfrc_con, 0, sizeof d->qfrc_con);
        d->cost = 0;
        return;
    }
    /* warmstart: the cheaper of qacc_warmstart and qacc_smooth */
    if (warm && total_cost(d, nv, warm) <= total_cost(d, nv, d->qacc_smooth)) memcpy(d->qacc, warm, nv * sizeof(double));
    else memcpy(d->qacc, d->qacc_smooth, nv * sizeof(double));

    double Ma[NV], Jaref[DLO_MAXEFC], Mv[NV], Jv[DLO_MAXEFC], grad[NV], Mgrad[NV], search[NV], gauss;
    double H[NV][NV], L[NV][NV];
    mul_M(d, nv, d->qacc, Ma);
    mul_J(d, nv, d->qacc, Jaref);
    for (int i = 0; i < nefc; i++) Jaref[i] -= d->earef[i];
    solver_update(d, nv, Ma, Jaref, &gauss, H, L, grad, Mgrad);
    for (int i = 0; i < nv; i++) search[i] = -Mgrad[i];
    double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
    int iter = 0;
    while (iter < m->iterations) {
        double snorm = 0;
        for (int i = 0; i < nv; i++) snorm += search[i] * search[i];
        snorm = sqrt(snorm);
        if (snorm < MINVAL) break;
        double gtol = m->tolerance * m->ls_tolerance * snorm * m->meaninertia * (nv > 1 ? nv : 1);
        mul_M(d, nv, search, Mv);
        mul_J(d, nv, search, Jv);
        lsctx c = {nv, nefc, d, gauss, 0, 0, Jaref, Jv};
        for (int i = 0; i < nv; i++) { c.g1 += search[i] * (Ma[i] - d->smooth[i]); c.g2 += 0.5 * search[i] * Mv[i]; }
        double alpha = linesearch(&c, gtol, m->ls_iterations);
        if (alpha == 0) break;
        for (int i = 0; i < nv; i++) { d->qacc[i] += alpha * search[i]; Ma[i] += alpha * Mv[i]; }
        for (int i = 0; i < nefc; i++) Jaref[i] += alpha * Jv[i];
        double oldcost = d->cost;
        solver_update(d, nv, Ma, Jaref, &gauss, H, L, grad, Mgrad);
        double gn = 0;
        for (int i = 0; i < nv; i++) gn += grad[i] * grad[i];
        double improvement = scale * (oldcost - d->cost), gradient = scale * sqrt(gn);
        iter++;
        if (improvement < m->tolerance || gradient < m->tolerance) break;
        for (int i = 0; i < nv; i++) search[i] = -Mgrad[i];
    }
    d->niter = iter;
}

/* [3P] mj_forward: position, velocity, actuation, acceleration and constraint stages */
static void forward(const model_t* mm, const double* q, const double* v, const double* ctrl, const double* warm, int flags, data_t* d) {
    const dl_model_desc* m = &mm->m;
    int nv = m->nv;
    kinematics(mm, q, d);
    mass_matrix(mm, d);
    chol_factor(nv, d->M, d->LM);
    collide(mm, d);
    if (flags & F_NOCONTACT) d->ncon = 0;
    make_constraint(mm, q, v, flags, d);
    bias_forces(mm, v, flags, d);
    for (int j = 0; j < nv; j++) d->passive[j] = (flags & F_NODAMP) ? 0 : -m->jnt_damping[j] * v[j];
    memset(d->actuator, 0, sizeof d->actuator);
    for (int a = 0; a < m->nu; a++) {
        double u = (flags & F_NOACT) || !ctrl ? 0 : ctrl[a];
        if (u < m->act_ctrlrange[a][0]) u = m->act_ctrlrange[a][0];
        if (u > m->act_ctrlrange[a][1]) u = m->act_ctrlrange[a][1];
        double f = u; /* motor: gain 1, no bias */
        if (f < m->act_forcerange[a][0]) f = m->act_forcerange[a][0];
        if (f > m->act_forcerange[a][1]) f = m->act_forcerange[a][1];
        d->act_force[a] = f;
        d->actuator[m->act_dof[a]] += m->act_gear[a] * f;
    }
    for (int j = 0; j < nv; j++) d->smooth[j] = d->passive[j] - d->bias[j] + d->actuator[j];
    if (mm->xfrc[0] != 0 || mm->xfrc[1] != 0 || mm->xfrc[2] != 0) {
        /* mj_xfrcAccumulate: qfrc_applied += Jp(com of body 1)^T F */
        double jp[3][NV], jr[3][NV];
        jac(mm, d, d->xipos[1], 1, jp, jr);
        for (int j = 0; j < nv; j++) d->smooth[j] += jp[0][j] * mm->xfrc[0] + jp[1][j] * mm->xfrc[1] + jp[2][j] * mm->xfrc[2];
    }
    chol_solve(nv, d->LM, d->smooth, d->qacc_smooth);
    solve(mm, warm, d);
}

static int bad(const double* x, int n) {
    for (int i = 0; i < n; i++)
        if (!(x[i] == x[i]) || x[i] > MAXVAL || x[i] < -MAXVAL) return 1;
    return 0;
}

/* Warm-start schedule of the RK4 stages.
 * 0 (default, what the device kernels do): every forward evaluation starts from and then overwrites qacc_warmstart, so
 *   stage k starts from the solution of stage k - 1.
 * 1 (MuJoCo 2.x as published: mj_forward never writes qacc_warmstart, mj_advance saves d->qacc once per mj_step): all four
 *   stages of a step start from the SAME vector, the one saved by the previous step -- which is the acceleration of that
 *   step's LAST stage, the last forward evaluation before mj_advance.
 * The minimiser of the convex solver cost is unique, so the two schedules differ in the iterate path only: results agree to
 * the solver tolerance (tests/test_oracle_physics.py::test_warmstart_schedules_agree).  Schedule 1 is what vectors dumped
 * from a real MuJoCo (tools/dump_mujoco_vectors.py -> tests/golden/G12_mujoco_step.npz) are compared under. */
static int g_warm_schedule = 0;
void dlo_set_warmstart_schedule(int schedule) { g_warm_schedule = schedule ? 1 : 0; }

/* [3P] mj_step with integrator RK4 (mj_RungeKutta, N = 4).  warm is qacc_warmstart (see the schedules above).
 * Returns 1 on divergence (mj_checkPos/Vel/Acc -> MujocoException in mujoco-py). */
static int mj_step_rk4(const model_t* mm, double* q, double* v, const double* ctrl, double* warm, double h, int flags, data_t* d) {
    int nv = mm->m.nv;
    static const double A[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 1}};
    static const double Bw[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
    double X[4][2 * NV], F[4][NV], warm_in[NV];
    const int per_step = g_warm_schedule == 1;
    if (bad(q, nv) || bad(v, nv)) return 1;
    memcpy(warm_in, warm, nv * sizeof(double));
    forward(mm, q, v, ctrl, warm, flags, d);
    if (!per_step) memcpy(warm, d->qacc, nv * sizeof(double));
    if (bad(d->qacc, nv)) return 1;
    memcpy(X[0], q, nv * sizeof(double));
    memcpy(X[0] + NV, v, nv * sizeof(double));
    memcpy(F[0], d->qacc, nv * sizeof(double));
    for (int i = 1; i < 4; i++) {
        for (int j = 0; j < nv; j++) {
            double dx = 0, df = 0;
            for (int k = 0; k < i; k++) { dx += A[i - 1][k] * X[k][NV + j]; df += A[i - 1][k] * F[k][j]; }
            X[i][j] = X[0][j] + h * dx;
            X[i][NV + j] = X[0][NV + j] + h * df;
        }
        forward(mm, X[i], X[i] + NV, ctrl, per_step ? warm_in : warm, flags, d);
        if (!per_step) memcpy(warm, d->qacc, nv * sizeof(double));
        memcpy(F[i], d->qacc, nv * sizeof(double));
    }
    if (per_step) memcpy(warm, d->qacc, nv * sizeof(double));       /* mj_advance: d->qacc is the last stage's */
    for (int j = 0; j < nv; j++) {
        double dx = 0, df = 0;
        for (int k = 0; k < 4; k++) { dx += Bw[k] * X[k][NV + j]; df += Bw[k] * F[k][j]; }
        q[j] = X[0][j] + h * dx;
        v[j] = X[0][NV + j] + h * df;
    }
    return 0;
}

/* ------------------------------------------------------------------ mj_setConst */
void dlo_set_const(dl_model_desc* m) {
    model_t mm;
    static data_t d;
    model_init(&mm, m);
    int nv = m->nv;
    double q0[NV] = {0};
    for (int j = 0; j < nv; j++) q0[j] = m->jnt_qpos0[j];
    kinematics(&mm, q0, &d);
    mass_matrix(&mm, &d);
    chol_factor(nv, d.M, d.LM);
    double Minv[NV][NV];
    for (int j = 0; j < nv; j++) {
        double e[NV] = {0}, x[NV];
        e[j] = 1;
        chol_solve(nv, d.LM, e, x);
        for (int i = 0; i < nv; i++) Minv[i][j] = x[i];
    }
    double tr = 0;
    for (int j = 0; j < nv; j++) { m->dof_invweight0[j] = Minv[j][j]; tr += d.M[j][j]; }
    m->meaninertia = tr / nv;
    m->body_invweight0[0][0] = m->body_invweight0[0][1] = 0;
    for (int b = 1; b < m->nbody; b++) {
        double jp[3][NV], jr[3][NV];
        jac(&mm, &d, d.xipos[b], b, jp, jr);
        double tp = 0, trr = 0;
        for (int k = 0; k < 3; k++)
            for (int i = 0; i < nv; i++)
                for (int j = 0; j < nv; j++) { tp += jp[k][i] * Minv[i][j] * jp[k][j]; trr += jr[k][i] * Minv[i][j] * jr[k][j]; }
        m->body_invweight0[b][0] = tp / 3;
        m->body_invweight0[b][1] = trr / 3;
    }
}

/* ------------------------------------------------------------------ RSI random stream */
static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
void dlo_rsi_draw(uint64_t seed, uint32_t global_env, uint32_t episode, int32_t n_steps, const int32_t* step_off, int32_t* i_step, int32_t* pos) {
    uint64_t r = splitmix64(seed ^ splitmix64(((uint64_t)global_env << 32) | episode));
    uint32_t lo = (uint32_t)r, hi = (uint32_t)(r >> 32);
    int32_t s = (int32_t)(((uint64_t)lo * (uint64_t)n_steps) >> 32);
    int32_t len = step_off[s + 1] - step_off[s];
    *i_step = s;
    *pos = (int32_t)(((uint64_t)hi * (uint64_t)len) >> 32);
}

/* ------------------------------------------------------------------ the vectorised environment */
typedef struct {
    double q[NV], v[NV], warm[NV];
    int32_t cur[DL_CUR_WORDS];
    double walked, comz_off;
    /* monitor (monitor_wrapper.py:88-133) */
    int32_t m_ep_len, m_has[8];
    double m_nsteps;
    double m_ret, m_last_rew, m_pos, m_vel, m_com, m_tor;
    double s_ep_len, s_ep_ret, s_mean_rew, s_pos, s_vel, s_com, s_tor, moved_distance;
    double pos_rew, vel_rew, com_rew, tor_abs_mean;
    int inject_exc, inject_state, inject_rsi, inj_rsi_step, inj_rsi_pos;
    double last_ctrl[DL_MAX_ACT];
    double inj_q[NV], inj_v[NV];
    /* build-defined dynamics randomisation (the reference's dynamics_randomization is a stub, mimic_env.py:492-524) */
    double mass_scale, floor_mu, xfrc[3];
    int randomized;
} walker_t;

struct dlo_env_s {
    model_t mm;
    dl_config cfg;
    int32_t n, n_steps, n_rows, total_len, stride;
    double* table;
    int32_t *step_off, *step_is_left;
    double* step_vel;
    walker_t* w;
    double* zacc; /* [n][n_steps]: quirk Q4, the COM-z offset every step of walker i's copy of the data set has accumulated (adjust_COM_Z_pos) */
    data_t d;
    int eval_mode;
};

dlo_env* dlo_create(const dl_model_desc* model, const dl_refs_desc* refs, const dl_config* cfg, int32_t n) {
    dlo_env* e = (dlo_env*)calloc(1, sizeof(dlo_env));
    model_init(&e->mm, model);
    e->cfg = *cfg;
    e->n = n;
    e->n_steps = refs->n_steps; e->n_rows = refs->n_rows; e->total_len = refs->total_len; e->stride = refs->stride;
    size_t tn = (size_t)refs->n_rows * refs->total_len;
    e->table = (double*)malloc(tn * sizeof(double));
    memcpy(e->table, refs->table, tn * sizeof(double));
    e->step_off = (int32_t*)malloc((refs->n_steps + 1) * sizeof(int32_t));
    memcpy(e->step_off, refs->step_off, (refs->n_steps + 1) * sizeof(int32_t));
    e->step_is_left = (int32_t*)malloc(refs->n_steps * sizeof(int32_t));
    memcpy(e->step_is_left, refs->step_is_left, refs->n_steps * sizeof(int32_t));
    e->step_vel = (double*)malloc(refs->n_steps * sizeof(double));
    memcpy(e->step_vel, refs->step_vel, refs->n_steps * sizeof(double));
    e->w = (walker_t*)calloc(n, sizeof(walker_t));
    e->zacc = (double*)calloc((size_t)n * refs->n_steps, sizeof(double));
    for (int i = 0; i < n; i++) {
        e->w[i].cur[DL_CUR_COUNT] = 1; /* straight_walk_trajecs.py:124 */
        for (int j = 0; j < model->nv; j++) e->w[i].q[j] = model->jnt_qpos0[j];
    }
    return e;
}
void dlo_destroy(dlo_env* e) {
    if (!e) return;
    free(e->table); free(e->step_off); free(e->step_is_left); free(e->step_vel); free(e->w); free(e->zacc); free(e);
}

static int step_len(const dlo_env* e, int s) { return e->step_off[s + 1] - e->step_off[s]; }
static int obs_dim(const dlo_env* e) { return e->cfg.env_kind == DL_ENV_LOCO3D ? 10 + 2 * e->mm.m.nv - 1 : 1 + 2 * e->mm.m.nv; }

/* refs.get_qpos()/get_qvel() at the cursor (base_ref_trajecs.py:44-56) incl. the COM-x offset of
 * _get_next_step (straight_walk_trajecs.py:338-347, quirk Q1) and the COM-z re-anchoring of
 * reset_model (mimic_env.py:555-557 -> adjust_COM_Z_pos, base_ref_trajecs.py:126-127).  Quirk Q4 (default): the
 * re-anchoring subtracts the offset from the COM-z row of the data set IN PLACE -- `_qpos_full` is `data[i_step]` itself after
 * an RSI reset (straight_walk_trajecs.py:466-468) and `data[0]` after an evaluation init (:242, quirk Q3) --, and the copy
 * `_get_next_step` makes at a rollover (:342) is a copy of the mutated row: whatever step the cursor reads carries the offsets of
 * all earlier resets of THIS environment that landed on it (every SubprocVecEnv worker has its own data set).  zacc[i][step] is
 * that sum.  With DL_INTENDED_COMZ_PER_EPISODE the offset applies to the reset step for the current episode only. */
static int q4_on(const dlo_env* e) { return !(e->cfg.intended_semantics & DL_INTENDED_COMZ_PER_EPISODE); }
static void ref_lookup(const dlo_env* e, const walker_t* w, double* qr, double* vr) {
    int nv = e->mm.m.nv;
    int base = e->step_off[w->cur[DL_CUR_READ_STEP]] + w->cur[DL_CUR_POS];
    for (int j = 0; j < nv; j++) {
        qr[j] = e->table[(size_t)j * e->total_len + base];
        vr[j] = e->table[(size_t)(nv + j) * e->total_len + base];
    }
    if (q4_on(e)) qr[2] -= e->zacc[(size_t)(w - e->w) * e->n_steps + w->cur[DL_CUR_READ_STEP]];
    if (w->cur[DL_CUR_HAS_DIST]) {
        int rs = w->cur[DL_CUR_RSI_STEP];
        qr[0] += e->table[e->step_off[rs + 1] - 1];
    } else if (!q4_on(e)) {
        qr[2] -= w->comz_off;
    }
}
void dlo_ref_lookup(dlo_env* e, int32_t i, double* qref, double* vref) { ref_lookup(e, &e->w[i], qref, vref); }

/* StraightWalkingTrajectories.next (straight_walk_trajecs.py:141-159) + _get_next_step (:322-348) */
static void cursor_next(const dlo_env* e, walker_t* w) {
    int32_t* c = w->cur;
    if (e->cfg.env_kind == DL_ENV_LOCO3D) {
        /* BaseReferenceTrajectories.next (base_ref_trajecs.py:95-103): the cursor wraps to 0 */
        c[DL_CUR_POS] += e->stride;
        if (c[DL_CUR_POS] >= step_len(e, 0) - 1) c[DL_CUR_POS] = 0;
        return;
    }
    c[DL_CUR_POS] += e->stride;
    int dif = c[DL_CUR_POS] - step_len(e, c[DL_CUR_READ_STEP]) + 1;
    if (dif > 0) {
        if (c[DL_CUR_I_STEP] >= e->n_steps - 1) c[DL_CUR_I_STEP] = e->step_is_left[c[DL_CUR_I_STEP]] ? 0 : 1;
        else { c[DL_CUR_I_STEP] += 1; c[DL_CUR_COUNT] += 1; }
        c[DL_CUR_HAS_DIST] = 1;
        c[DL_CUR_READ_STEP] = c[DL_CUR_I_STEP];
        c[DL_CUR_POS] = dif;
    }
}

/* MimicEnv._get_obs (mimic_env.py:403-437) + mirror_obs (:440-480) */
static double np_pairwise(const double* a, int n);
static void get_obs(const dlo_env* e, const walker_t* w, double* o) {
    int nv = e->mm.m.nv;
    const int32_t* c = w->cur;
    if (e->cfg.env_kind == DL_ENV_LOCO3D) {
        /* estimate_phase_vars_from_joint_phase_plots (mimic_env.py:330-401) on the joints
         * [9,12,14,17] (mimic_walker_165cm_65kg.py:40-43): angle atan2(v,-q)/pi and |(q,v)|/5 */
        static const int pj[4] = {9, 12, 14, 17};
        for (int k = 0; k < 4; k++) {
            double q = w->q[pj[k]], v = w->v[pj[k]];
            o[2 * k] = atan2(v, -q) / M_PI;
            o[2 * k + 1] = sqrt(q * q + v * v) / 5;
        }
        /* Loco3dReferenceTrajectories.get_desired_walking_velocity_vector (loco3d_trajecs.py:51-97):
         * mean of the reference pelvis x / z velocities over the next 0.5 s (clipped at the end) */
        int L = step_len(e, 0), pos = c[DL_CUR_POS], nt = (int)(0.5 * 500);
        int end = pos + nt < L - 1 ? pos + nt : L - 1;
        for (int r = 0; r < 2; r++) {
            const double* row = e->table + (size_t)(nv + r) * e->total_len;
            /* np.mean of a float64 slice: pairwise summation */
            int n = end - pos;
            double sum;
            if (n <= 0) sum = 0.0 / 0.0; else sum = np_pairwise(row + pos, n);
            o[8 + r] = sum / (n > 0 ? n : 1);
            if (n <= 0) o[8 + r] = 0.0 / 0.0;
        }
        for (int j = 1; j < nv; j++) o[9 + j] = w->q[j];
        for (int j = 0; j < nv; j++) o[9 + nv + j] = w->v[j];
        return;
    }
    o[0] = (double)c[DL_CUR_POS] / (double)step_len(e, c[DL_CUR_READ_STEP]);
    int iv = c[DL_CUR_I_STEP] - c[DL_CUR_COUNT] + 1;
    o[1] = e->step_vel[iv > 0 ? iv : 0];
    for (int j = 1; j < nv; j++) o[1 + j] = w->q[j];
    for (int j = 0; j < nv; j++) o[1 + nv + j] = w->v[j];
    if (e->cfg.mirror_policy && nv == 14 && e->step_is_left[c[DL_CUR_I_STEP]]) {
        static const int perm[29] = {0, 1, 2, 3, 4, 5, 6, 11, 12, 13, 14, 7, 8, 9, 10, 15, 16, 17, 18, 19, 20, 25, 26, 27, 28, 21, 22, 23, 24};
        static const int neg[10] = {2, 4, 6, 8, 12, 16, 18, 20, 22, 26};
        double t[29];
        for (int k = 0; k < 29; k++) t[k] = o[perm[k]];
        for (int k = 0; k < 10; k++) t[neg[k]] = -t[neg[k]];
        memcpy(o, t, sizeof t);
    }
}

/* np.sum of a contiguous float64 vector with n <= 128: numpy's pairwise kernel (8 partial sums
 * combined as a tree, then the tail), so that the reward is bit-identical to the reference's */
static double np_sum(const double* a, int n) {
    double res = 0;
    if (n < 8) {
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r[8];
    int i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; k++) r[k] += a[i + k];
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* numpy's pairwise summation for any n (np.mean of a contiguous float64 slice) */
static double np_pairwise(const double* a, int n) {
    if (n <= 128) return np_sum(a, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}

/* get_imitation_reward (mimic_env.py:592-649) */
static double imitation_reward(const dlo_env* e, walker_t* w) {
    int nv = e->mm.m.nv;
    double qr[NV], vr[NV], dp[NV], dv[NV], dc[3];
    ref_lookup(e, w, qr, vr);
    for (int j = 3; j < nv; j++) { dp[j - 3] = (w->q[j] - qr[j]) * (w->q[j] - qr[j]); dv[j - 3] = (w->v[j] - vr[j]) * (w->v[j] - vr[j]); }
    for (int j = 0; j < 3; j++) dc[j] = (w->q[j] - qr[j]) * (w->q[j] - qr[j]);
    double sp = np_sum(dp, nv - 3), sv = np_sum(dv, nv - 3), sc = np_sum(dc, 3);
    w->pos_rew = exp(-3 * sp);
    w->vel_rew = exp(-0.05 * sv);
    w->com_rew = exp(-16 * sc);
    return (e->cfg.rew_weights[0] * w->pos_rew + e->cfg.rew_weights[1] * w->vel_rew + e->cfg.rew_weights[2] * w->com_rew) * e->cfg.rew_scale;
}

static double lowest_site(dlo_env* e, const double* q) {
    const dl_model_desc* m = &e->mm.m;
    kinematics(&e->mm, q, &e->d);
    double low = 1e300;
    for (int s = 0; s < m->nsite; s++) {
        int b = m->site_body[s];
        double t[3];
        m3_mulv(t, e->d.xmat[b], m->site_pos[s]);
        double z = e->d.xpos[b][2] + t[2];
        if (z < low) low = z;
    }
    return low;
}

/* the model a walker is simulated with: the shared one, or a copy with its body masses / inertias scaled, its own
 * floor friction and its current push force */
static const model_t* walker_model(const dlo_env* e, const walker_t* w, model_t* tmp) {
    if (!w->randomized) return &e->mm;
    *tmp = e->mm;
    for (int b = 1; b < tmp->m.nbody; b++) {
        tmp->m.body_mass[b] *= w->mass_scale;
        for (int k = 0; k < 3; k++) tmp->m.body_inertia[b][k] *= w->mass_scale;
    }
    tmp->m.floor_friction = w->floor_mu;
    memcpy(tmp->xfrc, w->xfrc, sizeof tmp->xfrc);
    return tmp;
}

/* MujocoEnv.reset + MimicEnv.reset_model (mimic_env.py:526-572), RSI (straight_walk_trajecs.py:460-474) */
static void reset_walker(dlo_env* e, int i, int inj_step, int inj_pos, double* obs) {
    walker_t* w = &e->w[i];
    const dl_model_desc* m = &e->mm.m;
    int32_t* c = w->cur;
    int32_t s, p;
    int read = -1;
    if (inj_step >= 0) { s = inj_step; p = inj_pos; }
    else if (e->eval_mode && e->cfg.env_kind == DL_ENV_LOCO3D) {
        s = 0; p = 0;      /* base get_deterministic_init_state(pos_in_percent=0), base_ref_trajecs.py:66-73 */
    }
    else if (e->eval_mode) {
        /* _get_deterministic_init_state (straight_walk_trajecs.py:237-265): step k, 75 % of ITS length,
         * but the kinematics are read from step 0's table (quirk Q3) */
        s = c[DL_CUR_EVAL_K];
        p = (int32_t)(0.75 * step_len(e, s));
        read = (e->cfg.intended_semantics & DL_INTENDED_EVAL_OWN_STEP) ? s : 0;
        c[DL_CUR_EVAL_K] = (s + 1 >= 20) ? 0 : s + 1;
    }
    else if (w->inject_rsi) { s = w->inj_rsi_step; p = w->inj_rsi_pos; }
    else dlo_rsi_draw(e->cfg.seed, (uint32_t)(e->cfg.env_index_base + i), (uint32_t)c[DL_CUR_EPISODE], e->n_steps, e->step_off, &s, &p);
    c[DL_CUR_EPISODE] += 1;
    c[DL_CUR_EP_DUR] = 0;
    w->walked = 0;
    c[DL_CUR_I_STEP] = s; c[DL_CUR_RSI_STEP] = s; c[DL_CUR_READ_STEP] = read >= 0 ? read : s; c[DL_CUR_POS] = p; c[DL_CUR_HAS_DIST] = 0;
    if (e->cfg.intended_semantics & DL_INTENDED_COUNT_PER_EPISODE) c[DL_CUR_COUNT] = 1;
    w->comz_off = 0;
    ref_lookup(e, w, w->q, w->v);          /* Q4: the state is read from the row as earlier resets left it ... */
    double low = lowest_site(e, w->q);
    w->q[2] -= low;                        /* ... so the initial state itself does not depend on the history (the foot height follows the root's z) */
    w->comz_off = low;
    if (q4_on(e)) e->zacc[(size_t)i * e->n_steps + c[DL_CUR_READ_STEP]] += low;          /* adjust_COM_Z_pos: `row -= low`, in place */
    /* set_state -> mj_forward: qacc of the initial state becomes the warmstart */
    { model_t tmp; forward(walker_model(e, w, &tmp), w->q, w->v, NULL, NULL, 0, &e->d); }
    memcpy(w->warm, e->d.qacc, m->nv * sizeof(double));
    /* :562 get_imitation_reward() for the sanity assert leaves the components at exactly 1 */
    w->pos_rew = w->vel_rew = w->com_rew = 1.0;
    cursor_next(e, w);
    if (obs) get_obs(e, w, obs);
}

static void smooth(double* state, int32_t* has, double x, double alpha) {
    /* exponential_running_smoothing, drloco/common/utils.py:312-329 */
    if (!*has) { *state = x; *has = 1; }
    else *state = alpha * x + (1 - alpha) * *state;
}

/* Monitor.step bookkeeping (monitor_wrapper.py:88-133) */
static void monitor_step(walker_t* w, double rew, int done) {
    w->m_ep_len += 1;
    w->m_nsteps += 1;
    w->m_ret += rew;
    w->m_last_rew = rew;
    w->m_pos += w->pos_rew; w->m_vel += w->vel_rew; w->m_com += w->com_rew;
    w->m_tor += w->tor_abs_mean;
    if (done) {
        double n = (double)w->m_ep_len;
        smooth(&w->s_mean_rew, &w->m_has[0], (w->m_ret - w->m_last_rew) / (n - 1), 0.9);
        /* ep_pos_rews / ep_vel_rews / ep_com_rews are never cleared in the reference
         * (monitor_wrapper.py:100-102,116-118): the component means run over the whole history */
        smooth(&w->s_pos, &w->m_has[1], w->m_pos / w->m_nsteps, 0.9);
        smooth(&w->s_vel, &w->m_has[2], w->m_vel / w->m_nsteps, 0.9);
        smooth(&w->s_com, &w->m_has[3], w->m_com / w->m_nsteps, 0.9);
        smooth(&w->s_ep_ret, &w->m_has[4], w->m_ret, 0.25);
        smooth(&w->s_ep_len, &w->m_has[5], n, 0.75);
        smooth(&w->s_tor, &w->m_has[6], w->m_tor / n, 0.75);
        w->moved_distance = w->walked;
        w->m_ep_len = 0;
        w->m_ret = 0;
        w->m_tor = 0;
    }
}

void dlo_reset(dlo_env* e, const uint8_t* mask, const int32_t* init_step, const int32_t* init_pos, double* obs_out) {
    int od = obs_dim(e);
    for (int i = 0; i < e->n; i++) {
        if (mask && !mask[i]) continue;
        reset_walker(e, i, init_step ? init_step[i] : -1, init_pos ? init_pos[i] : -1, obs_out ? obs_out + (size_t)i * od : NULL);
    }
}

/* MimicEnv.step (mimic_env.py:60-126) + vec-env auto reset */
void dlo_step(dlo_env* e, const double* actions, double* obs, double* rew, uint8_t* done, double* term_obs, double* rew_terms) {
    const dl_model_desc* m = &e->mm.m;
    int nv = m->nv, nu = m->nu, od = obs_dim(e);
    for (int i = 0; i < e->n; i++) {
        walker_t* w = &e->w[i];
        int32_t* c = w->cur;
        double ctrl[DL_MAX_ACT], o[64];
        /* _rescale_actions (:170-192): clip to [-1,1]; a>0 ? a*high : |a|*low */
        for (int a = 0; a < nu; a++) {
            double x = actions[(size_t)i * nu + a];
            x = x < -1 ? -1 : (x > 1 ? 1 : x);
            ctrl[a] = x > 0 ? x * m->act_ctrlrange[a][1] : fabs(x) * m->act_ctrlrange[a][0];
        }
        /* mirror_action (:483-489), evaluated with the cursor BEFORE refs.next() */
        if (e->cfg.mirror_policy && nu == 8 && e->step_is_left[c[DL_CUR_I_STEP]]) {
            static const int perm[8] = {4, 5, 6, 7, 0, 1, 2, 3};
            double t[8];
            for (int a = 0; a < 8; a++) t[a] = ctrl[perm[a]];
            t[1] = -t[1]; t[5] = -t[5];
            memcpy(ctrl, t, sizeof t);
        }
        memcpy(w->last_ctrl, ctrl, nu * sizeof(double));
        int exc = 0;
        if (w->inject_exc) { exc = 1; w->inject_exc = 0; }
        else if (w->inject_state) {
            memcpy(w->q, w->inj_q, nv * sizeof(double));
            memcpy(w->v, w->inj_v, nv * sizeof(double));
            w->inject_state = 0;
        } else {
            model_t tmp;
            const model_t* mw = walker_model(e, w, &tmp);
            for (int k = 0; k < m->frame_skip && !exc; k++) exc = mj_step_rk4(mw, w->q, w->v, ctrl, w->warm, m->timestep, 0, &e->d);
        }
        double tor = 0;
        for (int a = 0; a < nu; a++) {
            double f = ctrl[a];
            f = f < m->act_forcerange[a][0] ? m->act_forcerange[a][0] : (f > m->act_forcerange[a][1] ? m->act_forcerange[a][1] : f);
            tor += fabs(f);
        }
        w->tor_abs_mean = tor / nu;
        double r;
        int dn;
        if (exc) {
            /* :86-91  obs = self.reset(); return obs, 0, True, {} */
            reset_walker(e, i, -1, -1, o);
            r = 0; dn = 1;
        } else {
            cursor_next(e, w);                          /* :96 */
            get_obs(e, w, o);                           /* :99 */
            c[DL_CUR_EP_DUR] += 1;                      /* :106 */
            double vx = w->v[0], vy = w->v[1];          /* :131-139 */
            vx = vx < -5.5 ? -5.5 : (vx > 5.5 ? 5.5 : vx);
            vy = vy < -5.5 ? -5.5 : (vy > 5.5 ? 5.5 : vy);
            w->walked += sqrt(vx * vx + vy * vy) * 1 / e->cfg.ctrl_freq;
            dn = (w->q[2] < e->cfg.com_z_min) || (c[DL_CUR_EP_DUR] >= e->cfg.ep_dur_max); /* :113-120 */
            /* :124,:142-168  _get_ET_reward is 0 (falls: -1*0 = -0.0) because ep_rews stays empty */
            if (dn) r = (c[DL_CUR_EP_DUR] >= e->cfg.ep_dur_max) ? 0.0 : -0.0;
            else r = imitation_reward(e, w) + e->cfg.alive_bonus;
        }
        monitor_step(w, r, dn);
        if (rew_terms) { rew_terms[3 * i] = w->pos_rew; rew_terms[3 * i + 1] = w->vel_rew; rew_terms[3 * i + 2] = w->com_rew; }
        rew[i] = r;
        done[i] = (uint8_t)dn;
        if (dn) {
            if (term_obs) memcpy(term_obs + (size_t)i * od, o, od * sizeof(double));
            reset_walker(e, i, -1, -1, o);
        }
        memcpy(obs + (size_t)i * od, o, od * sizeof(double));
    }
}

void dlo_get_state(dlo_env* e, double* qpos, double* qvel, double* qacc_warm, int32_t* cursor, double* walked) {
    int nv = e->mm.m.nv, n = e->n;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < nv; j++) {
            if (qpos) qpos[(size_t)j * n + i] = e->w[i].q[j];
            if (qvel) qvel[(size_t)j * n + i] = e->w[i].v[j];
            if (qacc_warm) qacc_warm[(size_t)j * n + i] = e->w[i].warm[j];
        }
        if (cursor) for (int k = 0; k < DL_CUR_WORDS; k++) cursor[(size_t)k * n + i] = e->w[i].cur[k];
        if (walked) walked[i] = e->w[i].walked;
    }
}
void dlo_set_state(dlo_env* e, const double* qpos, const double* qvel, const double* qacc_warm, const int32_t* cursor, const double* walked) {
    int nv = e->mm.m.nv, n = e->n;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < nv; j++) {
            if (qpos) e->w[i].q[j] = qpos[(size_t)j * n + i];
            if (qvel) e->w[i].v[j] = qvel[(size_t)j * n + i];
            if (qacc_warm) e->w[i].warm[j] = qacc_warm[(size_t)j * n + i];
        }
        if (cursor) for (int k = 0; k < DL_CUR_WORDS; k++) e->w[i].cur[k] = cursor[(size_t)k * n + i];
        if (walked) e->w[i].walked = walked[i];
    }
}
/* quirk Q4's record in the layout of dl_get_ref_offsets: double[n_steps, N] */
void dlo_get_ref_offsets(dlo_env* e, double* z) {
    for (int i = 0; i < e->n; i++) for (int s = 0; s < e->n_steps; s++) z[(size_t)s * e->n + i] = e->zacc[(size_t)i * e->n_steps + s];
}
void dlo_set_ref_offsets(dlo_env* e, const double* z) {
    for (int i = 0; i < e->n; i++) for (int s = 0; s < e->n_steps; s++) e->zacc[(size_t)i * e->n_steps + s] = z[(size_t)s * e->n + i];
}
void dlo_forward(dlo_env* e, const double* ctrl, double* qacc, int32_t* ncon, int32_t* nefc, int32_t* niter) {
    int nv = e->mm.m.nv, nu = e->mm.m.nu, n = e->n;
    for (int i = 0; i < n; i++) {
        double u[DL_MAX_ACT];
        for (int a = 0; a < nu; a++) u[a] = ctrl ? ctrl[(size_t)a * n + i] : 0;
        { model_t tmp; forward(walker_model(e, &e->w[i], &tmp), e->w[i].q, e->w[i].v, u, e->w[i].warm, 0, &e->d); }
        for (int j = 0; j < nv; j++) qacc[(size_t)j * n + i] = e->d.qacc[j];
        if (ncon) ncon[i] = e->d.ncon;
        if (nefc) nefc[i] = e->d.nefc;
        if (niter) niter[i] = e->d.niter;
    }
}
void dlo_set_eval(dlo_env* e, int32_t on) { e->eval_mode = on != 0; }
void dlo_inject_exception(dlo_env* e, int32_t i) { e->w[i].inject_exc = 1; }
void dlo_inject_rsi(dlo_env* e, int32_t i, int32_t step, int32_t pos) {
    e->w[i].inject_rsi = step >= 0;
    e->w[i].inj_rsi_step = step;
    e->w[i].inj_rsi_pos = pos;
}
void dlo_inject_state(dlo_env* e, int32_t i, const double* qpos, const double* qvel) {
    e->w[i].inject_state = 1;
    memcpy(e->w[i].inj_q, qpos, e->mm.m.nv * sizeof(double));
    memcpy(e->w[i].inj_v, qvel, e->mm.m.nv * sizeof(double));
}

/* test hooks */
void dlo_observe(dlo_env* e, double* obs, double* imit, double* terms) {
    int od = obs_dim(e);
    for (int i = 0; i < e->n; i++) {
        walker_t* w = &e->w[i];
        get_obs(e, w, obs + (size_t)i * od);
        imit[i] = imitation_reward(e, w);
        terms[3 * i] = w->pos_rew; terms[3 * i + 1] = w->vel_rew; terms[3 * i + 2] = w->com_rew;
    }
}
void dlo_last_ctrl(dlo_env* e, double* out) {
    for (int i = 0; i < e->n; i++)
        for (int a = 0; a < e->mm.m.nu; a++) out[(size_t)i * e->mm.m.nu + a] = e->w[i].last_ctrl[a];
}
void dlo_monitor_feed(dlo_env* e, int32_t i, double rew, int32_t done, double pos, double vel, double com, double tor, double walked) {
    walker_t* w = &e->w[i];
    w->pos_rew = pos; w->vel_rew = vel; w->com_rew = com; w->tor_abs_mean = tor; w->walked = walked;
    monitor_step(w, rew, done);
}

int dlo_stats_snapshot(dlo_env* e, const char* name, double* out) {
    for (int i = 0; i < e->n; i++) {
        const walker_t* w = &e->w[i];
        if (!strcmp(name, "ep_len_smoothed")) out[i] = w->s_ep_len;
        else if (!strcmp(name, "ep_ret_smoothed")) out[i] = w->s_ep_ret;
        else if (!strcmp(name, "mean_reward_smoothed")) out[i] = w->s_mean_rew;
        else if (!strcmp(name, "moved_distance")) out[i] = w->moved_distance;
        else if (!strcmp(name, "mean_ep_pos_rew_smoothed")) out[i] = w->s_pos;
        else if (!strcmp(name, "mean_ep_vel_rew_smoothed")) out[i] = w->s_vel;
        else if (!strcmp(name, "mean_ep_com_rew_smoothed")) out[i] = w->s_com;
        else if (!strcmp(name, "mean_abs_ep_torque_smoothed")) out[i] = w->s_tor;
        else return -1;
    }
    return 0;
}

/* per-walker dynamics randomisation / push force; NULL leaves a field unchanged */
void dlo_set_randomization(dlo_env* e, const double* mass_scale, const double* floor_friction, const double* xfrc) {
    for (int i = 0; i < e->n; i++) {
        walker_t* w = &e->w[i];
        if (!w->randomized) { w->mass_scale = 1.0; w->floor_mu = e->mm.m.floor_friction; w->xfrc[0] = w->xfrc[1] = w->xfrc[2] = 0; w->randomized = 1; }
        if (mass_scale) w->mass_scale = mass_scale[i];
        if (floor_friction) w->floor_mu = floor_friction[i];
        if (xfrc) memcpy(w->xfrc, xfrc + 3 * (size_t)i, 3 * sizeof(double));
    }
}

/* do_terminate_early (mimic_env.py:652-702): [any, com height, trunk angle, com-y] */
void dlo_terminate_early(dlo_env* e, int32_t i, int32_t* flags) {
    walker_t* w = &e->w[i];
    double qr[NV], vr[NV];
    ref_lookup(e, w, qr, vr);
    int low = w->q[2] < 0.75;
    int drunk = fabs(w->q[1]) > 0.2;
    int front = fabs(w->q[3] - qr[3]) > 0.2;
    int sag = w->q[4] > 0.3 || w->q[4] < -0.05;
    int trunk = sag || front;
    flags[0] = low || trunk || drunk; flags[1] = low; flags[2] = trunk; flags[3] = drunk;
}

/* ------------------------------------------------------------------ physics probes */
void dlo_probe_forward(const dl_model_desc* m, const double* qpos, const double* qvel, const double* ctrl, const double* warm, int flags, dlo_probe* out) {
    static model_t mm;
    static data_t d;
    model_init(&mm, m);
    int nv = m->nv;
    forward(&mm, qpos, qvel, ctrl, warm, flags, &d);
    memset(out, 0, sizeof *out);
    for (int i = 0; i < nv; i++) {
        for (int j = 0; j < nv; j++) out->M[i * nv + j] = d.M[i][j];
        out->qfrc_bias[i] = d.bias[i]; out->qfrc_smooth[i] = d.smooth[i]; out->qacc_smooth[i] = d.qacc_smooth[i];
        out->qacc[i] = d.qacc[i]; out->qfrc_constraint[i] = d.qfrc_con[i];
    }
    for (int b = 0; b < m->nbody; b++) {
        memcpy(out->xpos + 3 * b, d.xpos[b], 3 * sizeof(double));
        memcpy(out->xmat + 9 * b, d.xmat[b], 9 * sizeof(double));
        memcpy(out->xipos + 3 * b, d.xipos[b], 3 * sizeof(double));
    }
    for (int s = 0; s < m->nsite; s++) {
        double t[3];
        m3_mulv(t, d.xmat[m->site_body[s]], m->site_pos[s]);
        v3_add(out->site_xpos + 3 * s, d.xpos[m->site_body[s]], t);
    }
    double pe = 0, ke = 0;
    for (int b = 1; b < m->nbody; b++) pe -= m->body_mass[b] * v3_dot(m->gravity, d.xipos[b]);
    for (int i = 0; i < nv; i++)
        for (int j = 0; j < nv; j++) ke += 0.5 * qvel[i] * d.M[i][j] * qvel[j];
    out->energy[0] = pe; out->energy[1] = ke;
    out->ncon = d.ncon; out->nefc = d.nefc; out->niter = d.niter; out->solver_cost = d.cost;
    for (int c = 0; c < d.ncon; c++) {
        memcpy(out->con_pos + 3 * c, d.cpos[c], 3 * sizeof(double));
        memcpy(out->con_frame + 9 * c, d.cframe[c], 9 * sizeof(double));
        out->con_dist[c] = d.cdist[c];
        out->con_geom[c] = d.cgeom[c];
    }
    for (int i = 0; i < d.nefc; i++) {
        for (int j = 0; j < nv; j++) out->efc_J[i * nv + j] = d.J[i][j];
        out->efc_pos[i] = d.epos[i]; out->efc_D[i] = d.eD[i]; out->efc_aref[i] = d.earef[i]; out->efc_force[i] = d.eforce[i];
    }
}

int dlo_probe_steps(const dl_model_desc* m, double* qpos, double* qvel, const double* ctrl, double* warm, double dt, int n, int flags) {
    static model_t mm;
    static data_t d;
    model_init(&mm, m);
    mm.m.timestep = dt;
    for (int k = 0; k < n; k++)
        if (mj_step_rk4(&mm, qpos, qvel, ctrl, warm, dt, flags, &d)) return k + 1;
    return 0;
}

/* ------------------------------------------------------------------ SB3 reductions */
/* RunningMeanStd.update_from_moments (SB3 1.0): population variance of the batch, Chan merge */
void dlo_moments_update(double* mean, double* var, double* count, const double* x, int32_t B, int32_t D) {
    for (int k = 0; k < D; k++) {
        double bm = 0, bv = 0;
        for (int i = 0; i < B; i++) bm += x[(size_t)i * D + k];
        bm /= B;
        for (int i = 0; i < B; i++) bv += (x[(size_t)i * D + k] - bm) * (x[(size_t)i * D + k] - bm);
        bv /= B;
        double delta = bm - mean[k], tot = *count + B;
        double m_a = var[k] * *count, m_b = bv * B;
        double M2 = m_a + m_b + delta * delta * *count * B / tot;
        mean[k] = mean[k] + delta * B / tot;
        var[k] = M2 / tot;
    }
    *count += B;
}

/* RolloutBuffer.compute_returns_and_advantage (SB3 1.0), float32 like the numpy buffers */
void dlo_gae(const float* rew, const float* val, const uint8_t* ep_start, const float* last_val, const uint8_t* last_done, float gamma, float lam, int32_t T, int32_t N, float* adv, float* ret) {
    for (int i = 0; i < N; i++) {
        float last = 0;
        for (int t = T - 1; t >= 0; t--) {
            float nnt, nv;
            if (t == T - 1) { nnt = 1.0f - (float)last_done[i]; nv = last_val[i]; }
            else { nnt = 1.0f - (float)ep_start[(size_t)(t + 1) * N + i]; nv = val[(size_t)(t + 1) * N + i]; }
            float delta = rew[(size_t)t * N + i] + gamma * nv * nnt - val[(size_t)t * N + i];
            last = delta + gamma * lam * nnt * last;
            adv[(size_t)t * N + i] = last;
            ret[(size_t)t * N + i] = last + val[(size_t)t * N + i];
        }
    }
}
