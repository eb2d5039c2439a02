"""von-Mises-Fisher scaling layer 'f' on S2 with optional vertical ('r...') and azimuthal ('o...') spline flows -- host side
(jammy_flows/layers/spheres/fvm_2d.py:28-824).  This is the "2-sphere autoregressive spline" layer (legacy letter 'n').

Where the reference embeds nested ``pdf`` objects as passthrough flows (:193-197, 221-225) this implementation flattens them at
construction into the jf_f_layer descriptor; the whole layer (rotation, kappa map, nested splines, S2 <-> R2 chart) is ONE launch of
the 'f' HIP kernel (jf_f_chain_*).  The correlated variant (azimuthal spline parameters emitted per sample by an MLP of z whose weights are themselves in the
layer's row) is evaluated inside the same kernel.
"""
import torch
from torch import nn

from . import sphere_base
from .splines_1d import spline_1d
from ..intervals.rational_quadratic_spline import rational_quadratic_spline
from ... import _hip


class fisher_von_mises_2d(sphere_base.sphere_base):
    FAMILY = "f"

    def __init__(self, dimension, euclidean_to_sphere_as_first=False, use_permanent_parameters=False, fisher_parametrization="split",
                 add_vertical_rq_spline_flow=0, add_circular_rq_spline_flow=0, vertical_flow_defs="r", circular_flow_defs="o",
                 add_correlated_rq_spline_flow=0, correlated_max_rank=3, inverse_z_scaling=1, spline_num_basis_functions=5,
                 boundary_cos_theta_identity_region=0.0, vertical_smooth=0, vertical_restrict_max_min_width_height_ratio=-1.0,
                 vertical_fix_boundary_derivative=1, vertical_fix_first_width_n_height_to_zero=0, vertical_also_fix_second_width_to_zero=0,
                 vertical_independent_width_height_parametrization=0, circular_add_rotation=1, min_kappa=1e-10,
                 kappa_prediction="direct_log_real_bounded", add_extra_rotation_inbetween=0, kappa_clamping=0, add_rotation=1,
                 rotation_mode="householder", num_householder_iter=-1):
        """Symbol "f".  Parameters as in the reference (:58-80)."""
        super().__init__(dimension=dimension, euclidean_to_sphere_as_first=euclidean_to_sphere_as_first,
                         use_permanent_parameters=use_permanent_parameters, add_rotation=add_rotation,
                         num_householder_iter=num_householder_iter, rotation_mode=rotation_mode)
        if dimension != 2:
            raise Exception("2-D Flow")
        assert fisher_parametrization == "split"
        if kappa_prediction not in _hip.F_KAPPA_MODES:
            raise Exception("unknown kappa_prediction", kappa_prediction)
        self.z_scaling_factor = -1.0 if inverse_z_scaling else 1.0
        self.min_kappa = min_kappa
        self.kappa_prediction = kappa_prediction
        self.kappa_clamping = kappa_clamping
        # kappa from its own parameter (modes 0-2) or from the length of the rotation parameters (fvm_2d.py:105-139)
        self.kappa_fn = True if _hip.F_KAPPA_MODES[kappa_prediction] <= 2 else None
        if kappa_prediction in ("mu", "mu_squared"):
            assert self.add_rotation and self.rotation_mode == "xyz"
        if kappa_prediction in ("quatvec", "quatvec_squared"):
            assert self.add_rotation and self.rotation_mode == "quaternion", ("ROTATION MODE?!", self.rotation_mode)
        self.num_loglike_kappa_params = 0
        if self.kappa_fn is not None:
            self.num_loglike_kappa_params = 1
            if use_permanent_parameters:
                self.loglike_kappa = nn.Parameter(torch.randn(1).unsqueeze(0))
            self.total_param_num += 1
        self.add_vertical_rq_spline_flow = add_vertical_rq_spline_flow
        self.add_circular_rq_spline_flow = add_circular_rq_spline_flow
        self.add_correlated_rq_spline_flow = add_correlated_rq_spline_flow
        self.boundary_cos_theta_identity_region = boundary_cos_theta_identity_region
        self.spline_num_basis_functions = spline_num_basis_functions
        if spline_num_basis_functions == -1:
            assert vertical_smooth == 1, "num_basis_functions=-1 means alternating 2/3 as basis functions and requires smooth splines."

        # nested flows, flattened (the reference builds passthrough pdfs "i1_-b_b : r.." and "s1 : o.." here, :157-241)
        self._vertical, self._circular = [], []
        bound = float("%.2f" % (1.0 - boundary_cos_theta_identity_region))
        self.total_num_vertical_params = 0
        if add_vertical_rq_spline_flow:
            for i, letter in enumerate(vertical_flow_defs):
                assert letter == "r", "vertical flows must be 'r' layers"
                nb = spline_num_basis_functions if spline_num_basis_functions != -1 else (3 if i % 2 == 1 else 2)
                self._vertical.append(rational_quadratic_spline(
                    1, num_basis_functions=nb, euclidean_to_interval_as_first=0, use_permanent_parameters=0, low_boundary=-bound,
                    high_boundary=bound, fix_boundary_derivatives=-1.0 if vertical_fix_boundary_derivative == 0 else 1.0,
                    smooth_second_derivative=vertical_smooth, restrict_max_min_width_height_ratio=vertical_restrict_max_min_width_height_ratio,
                    fix_first_width_n_height_to_zero=vertical_fix_first_width_n_height_to_zero,
                    also_fix_second_width_to_zero=vertical_also_fix_second_width_to_zero,
                    independent_width_height_parametrization=vertical_independent_width_height_parametrization))
            self.total_num_vertical_params = sum(l.total_param_num for l in self._vertical)
            self.total_param_num += self.total_num_vertical_params
            if use_permanent_parameters:
                self.vertical_flow_params = nn.Parameter(torch.randn(1, self.total_num_vertical_params))
        self.total_num_circular_params = 0
        self.circular_add_rotation = circular_add_rotation
        if add_circular_rq_spline_flow:
            assert circular_add_rotation == 0, "Currently not allowing additional S-1 rotations due to potential complications at the poles."
            for letter in circular_flow_defs:
                assert letter == "o", "circular flows must be 'o' layers"
                self._circular.append(spline_1d(
                    1, euclidean_to_sphere_as_first=0, add_rotation=0, use_permanent_parameters=0, num_basis_functions=2, smooth_second_derivative=1,
                    fix_first_width_n_height_to_zero=vertical_fix_first_width_n_height_to_zero,
                    also_fix_second_width_to_zero=vertical_also_fix_second_width_to_zero,
                    independent_width_height_parametrization=vertical_independent_width_height_parametrization))
            self.total_num_circular_params = sum(l.total_param_num for l in self._circular)
            self.total_param_num += self.total_num_circular_params
            if use_permanent_parameters:
                self.circular_flow_params = nn.Parameter(torch.randn(1, self.total_num_circular_params))
        self.total_num_correlated_params = 0
        self._corr_mlp = None
        if add_correlated_rq_spline_flow:
            # the reference nests pdf("i1_-b_b+s1", vertical_flow_defs + "+" + circular_flow_defs, amortize_everything=True,
            # amortization_mlp_use_custom_mode=True, amortization_mlp_dims="64", amortization_mlp_ranks=correlated_max_rank) with DEFAULT
            # layer options (:244-262): the z splines take their rows from this layer's row, the azimuthal splines (incl. their own
            # Householder rotations) take theirs from a per-sample MLP of z whose weights are in this layer's row as well
            assert add_circular_rq_spline_flow == 0
            assert add_vertical_rq_spline_flow == 0
            from ... import flow_options
            from ...amortizable_mlp import AmortizableMLP
            r_opts, o_opts = flow_options.obtain_default_options("r"), flow_options.obtain_default_options("o")
            for letter in vertical_flow_defs:
                assert letter == "r", "vertical flows must be 'r' layers"
                self._vertical.append(rational_quadratic_spline(1, euclidean_to_interval_as_first=0, use_permanent_parameters=0,
                                                                low_boundary=-bound, high_boundary=bound, **r_opts))
            for letter in circular_flow_defs:
                assert letter == "o", "circular flows must be 'o' layers"
                self._circular.append(spline_1d(1, euclidean_to_sphere_as_first=0, use_permanent_parameters=0, **o_opts))
            n_vert = sum(l.total_param_num for l in self._vertical)
            n_circ = sum(l.total_param_num for l in self._circular)
            self._corr_mlp = AmortizableMLP(1, "64", n_circ, low_rank_approximations=correlated_max_rank, use_permanent_parameters=False)
            assert self._corr_mlp.stages[0]["full"] and len(self._corr_mlp.stages) == 2
            self.total_num_correlated_params = n_vert + self._corr_mlp.num_amortization_params
            self.total_param_num += self.total_num_correlated_params
            if n_circ + self._corr_mlp.stages[1]["rank"] > _hip.JF_CORR_SCRATCH - 1:
                raise NotImplementedError("correlated f flow: the nested circular flows need %d parameters, the kernel holds at most %d"
                                          % (n_circ, _hip.JF_CORR_SCRATCH - 1 - self._corr_mlp.stages[1]["rank"]))
            if use_permanent_parameters:
                self.correlated_flow_params = nn.Parameter(torch.randn(1, self.total_num_correlated_params))
        if len(self._vertical) > _hip.JF_MAX_NESTED or len(self._circular) > _hip.JF_MAX_NESTED:
            raise NotImplementedError("at most %d nested vertical / circular layers are supported by the kernel" % _hip.JF_MAX_NESTED)
        self.add_extra_rotation_inbetween = add_extra_rotation_inbetween

    def c_struct(self, first):
        L = _hip.jf_f_layer()
        L.hh_iter = self.num_householder_iter
        L.first = int(first)
        L.n_vertical = len(self._vertical)
        L.n_circular = len(self._circular)
        L.z_sign = float(self.z_scaling_factor)
        L.min_kappa = float(self.min_kappa)
        L.identity_region = float(self.boundary_cos_theta_identity_region)
        L.kappa_mode = _hip.F_KAPPA_MODES[self.kappa_prediction]
        L.kappa_clamping = 1 if self.kappa_clamping else 0
        L.extra_rotation = 1 if self.add_extra_rotation_inbetween else 0
        for i, l in enumerate(self._vertical):
            L.vertical[i] = l.c_struct(0)
        for i, l in enumerate(self._circular):
            L.circular[i] = l.c_struct(0)
        if self._corr_mlp is not None:
            st = self._corr_mlp.stages[1]
            L.correlated, L.corr_hidden, L.corr_rank, L.corr_full2 = 1, st["inp"], st["rank"], int(st["full"])
        return L

    def n_spline_calls(self):
        return len(self._vertical) + len(self._circular)

    def _layer_tensors(self):
        ts = [self.loglike_kappa] if self.kappa_fn is not None else []
        if self.add_vertical_rq_spline_flow:
            ts.append(self.vertical_flow_params)
        if self.add_circular_rq_spline_flow:
            ts.append(self.circular_flow_params)
        if self.add_correlated_rq_spline_flow:
            ts.append(self.correlated_flow_params)
        return ts

    def _init_params(self, params):
        c = self.num_loglike_kappa_params
        if self.kappa_fn is not None:
            self.loglike_kappa.data = params[:1].reshape(1, 1)
        assert len(params) == c + self.total_num_vertical_params + self.total_num_circular_params + self.total_num_correlated_params
        if self.add_correlated_rq_spline_flow:
            self.correlated_flow_params.data = params[c:c + self.total_num_correlated_params].reshape(1, -1)
        if self.add_vertical_rq_spline_flow:
            self.vertical_flow_params.data = params[c:c + self.total_num_vertical_params].reshape(1, -1)
            c += self.total_num_vertical_params
        if self.add_circular_rq_spline_flow:
            self.circular_flow_params.data = params[c:c + self.total_num_circular_params].reshape(1, -1)

    def _get_desired_init_parameters(self):
        parts = [torch.randn(1) - 3.0] if self.kappa_fn is not None else [torch.zeros(0)]       # log kappa (:750)
        parts += [l.get_desired_init_parameters() for l in self._vertical]
        if self._corr_mlp is not None:       # nested pdf.init_params() (:756-758): the MLP starts damped with the circular init as final bias
            circ = torch.cat([l.get_desired_init_parameters() for l in self._circular])
            parts.append(self._corr_mlp.obtain_default_init_tensor(fix_final_bias=circ).to(parts[0].dtype))
            return torch.cat(parts)
        parts += [l.get_desired_init_parameters() for l in self._circular]
        return torch.cat(parts)

    def _obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        if extra_inputs is not None:
            c = self.num_loglike_kappa_params
            if c:
                param_dict[extra_prefix + "loglike_kappa"] = extra_inputs[:, :1].data
            if self.add_vertical_rq_spline_flow:
                param_dict[extra_prefix + "vertical_params"] = extra_inputs[:, c:c + self.total_num_vertical_params].data
                c += self.total_num_vertical_params
            if self.add_circular_rq_spline_flow:
                param_dict[extra_prefix + "circular_params"] = extra_inputs[:, c:c + self.total_num_circular_params].data
            if self.add_correlated_rq_spline_flow:
                param_dict[extra_prefix + "correlated_params"] = extra_inputs[:, c:c + self.total_num_correlated_params].data
        else:
            if self.kappa_fn is not None:
                param_dict[extra_prefix + "loglike_kappa"] = self.loglike_kappa.data
            if self.add_vertical_rq_spline_flow:
                param_dict[extra_prefix + "vertical_params"] = self.vertical_flow_params.data
            if self.add_circular_rq_spline_flow:
                param_dict[extra_prefix + "circular_params"] = self.circular_flow_params.data
            if self.add_correlated_rq_spline_flow:
                param_dict[extra_prefix + "correlated_params"] = self.correlated_flow_params.data
