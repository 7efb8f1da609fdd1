// GP_Integrate: fourth-order Runge-Kutta integration of a system whose right-hand sides are
// BSS-ANOVA models taken from fits -- the consumer of (betas, mtx, phis) in the reference,
// /root/reference/src/FoKL/GP_Integrate.py:5-282.  Thousands of dependent steps with O(terms) work each: sequential
// and N-independent by nature, so it lives in the host half of the library.
//
// Restated semantics (GI = GP_Integrate.py):
//   * state j enters a model as clamp((y_j - min_j) / (max_j - min_j), 0, 1)                    GI:69-76
//   * model value = betas[0] + sum_i betas[i + 1] * prod_j spline_{mtx[i][j]}(x_j), splines on 498 intervals:
//     piece = floor(498 x) (497 at x = 1), local coordinate X = (x - piece / 498) / (1 / 498), value
//     c0 + c1 X + c2 X**2 + c3 X**3 summed left to right, the powers through libm pow as numpy does  GI:103-134
//   * the four stages, each multiplied by h, zeroed where the (intermediate) state sits on a bound of `norms` and the
//     slope points outwards                                                                     GI:204-269
//   * y += (dy1 + 2 dy2 + 2 dy3 + dy4) / 6                                                      GI:271
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/fokl_hip.h"

extern void fokl_set_global_error(const std::string &msg);   // fokl_hip.hip

namespace {

struct Model {
    const double *betas;
    const int32_t *mtx;      // [rows, cols] basis orders
    int rows, cols;
    const int32_t *source;   // [n_in]: >= 0 state index, < 0: -(c + 1) = column c of the forcing row
    int n_in;
};

inline double spline_value(const double *table, int width, int order, double x)
{
    int piece = (int)std::floor(x * 498.0);                 // GI:106 (the table has width = 499 pieces)
    if (piece == 498) piece -= 1;                           // GI:109-115
    const double r = 1.0 / 498.0;
    const double xmin = r * (double)piece;
    const double X = (x - xmin) / r;                        // GI:117-119
    const double *c = table + (size_t)(order - 1) * 4 * width;
    return c[piece] + c[width + piece] * X + c[2 * width + piece] * std::pow(X, 2.0) +
           c[3 * width + piece] * std::pow(X, 3.0);         // GI:128-131
}

}  // namespace

extern "C" int fokl_gp_integrate(int n_states, int n_other, int64_t n_steps, const double *const *betas,
                                 const int32_t *const *mtx, const int32_t *mtx_rows, const int32_t *mtx_cols,
                                 const int32_t *const *source, const int32_t *n_source, const double *forcing,
                                 const double *norms, const double *spline_table, int n_basis, int width, double h,
                                 double *y, double *trajectory)
{
    if (n_states <= 0 || n_other < 0 || n_steps < 0 || !betas || !mtx || !mtx_rows || !mtx_cols || !source ||
        !n_source || !norms || !spline_table || !y || !trajectory || width != 499 || (n_other > 0 && !forcing)) {
        fokl_set_global_error("fokl_gp_integrate: null pointer, empty system or a table that is not 499 pieces wide");
        return FOKL_ERR_ARG;
    }
    std::vector<Model> models((size_t)n_states);
    for (int k = 0; k < n_states; ++k) {
        models[k] = Model{betas[k], mtx[k], mtx_rows[k], mtx_cols[k], source[k], n_source[k]};
        if (models[k].n_in < models[k].cols) {
            fokl_set_global_error("fokl_gp_integrate: a model has more input columns than inputs are routed to it");
            return FOKL_ERR_ARG;
        }
        for (int i = 0; i < models[k].rows * models[k].cols; ++i)
            if (models[k].mtx[i] < 0 || models[k].mtx[i] > n_basis) {
                fokl_set_global_error("fokl_gp_integrate: basis order outside the spline table");
                return FOKL_ERR_ARG;
            }
        for (int i = 0; i < models[k].n_in; ++i) {
            const int s = models[k].source[i];
            if (s >= n_states || (s < 0 && -(s + 1) >= n_other)) {
                fokl_set_global_error("fokl_gp_integrate: input routing out of range");
                return FOKL_ERR_ARG;
            }
        }
    }
    const double *lo = norms, *hi = norms + n_states;       // norms [2, n_states]: minima, maxima
    const size_t stride = (size_t)n_steps + 1;
    std::vector<double> stage((size_t)n_states), dy1(stage), dy2(stage), dy3(stage), dy4(stage), x;
    for (int j = 0; j < n_states; ++j) trajectory[(size_t)j * stride] = y[j];

    auto slopes = [&](const double *at, const double *row, double *out) {
        for (int k = 0; k < n_states; ++k) {
            const Model &m = models[k];
            x.resize((size_t)m.n_in);
            for (int i = 0; i < m.n_in; ++i) {
                const int s = m.source[i];
                if (s >= 0) {
                    double v = (at[s] - lo[s]) / (hi[s] - lo[s]);
                    if (v > 1.0) v = 1.0;
                    if (v < 0.0) v = 0.0;
                    x[i] = v;
                } else {
                    x[i] = row[-(s + 1)];
                }
            }
            double delta = 0.0;
            for (int i = 0; i < m.rows; ++i) {
                double phi = 1.0;
                for (int j = 0; j < m.cols; ++j) {
                    const int order = m.mtx[i * m.cols + j];
                    if (order != 0) phi = phi * spline_value(spline_table, width, order, x[j]);
                }
                delta = delta + m.betas[i + 1] * phi;
            }
            out[k] = (delta + m.betas[0]) * h;
        }
    };
    auto saturate = [&](const double *at, double *dy) {
        for (int p = 0; p < n_states; ++p) {
            if (at[p] >= hi[p] && dy[p] > 0) dy[p] = 0;
            if (at[p] <= lo[p] && dy[p] < 0) dy[p] = 0;
        }
    };

    for (int64_t t = 0; t < n_steps; ++t) {
        const double *row = n_other > 0 ? forcing + (size_t)t * n_other : nullptr;
        slopes(y, row, dy1.data());
        saturate(y, dy1.data());
        for (int j = 0; j < n_states; ++j) stage[j] = y[j] + dy1[j] / 2;
        slopes(stage.data(), row, dy2.data());
        saturate(stage.data(), dy2.data());
        for (int j = 0; j < n_states; ++j) stage[j] = y[j] + dy2[j] / 2;
        slopes(stage.data(), row, dy3.data());
        saturate(stage.data(), dy3.data());
        for (int j = 0; j < n_states; ++j) stage[j] = y[j] + dy3[j];
        slopes(stage.data(), row, dy4.data());
        saturate(stage.data(), dy4.data());
        for (int j = 0; j < n_states; ++j) {
            y[j] += (dy1[j] + 2 * dy2[j] + 2 * dy3[j] + dy4[j]) / 6;
            trajectory[(size_t)j * stride + t + 1] = y[j];
        }
    }
    return FOKL_OK;
}
