"""Base class of spherical flow layers -- API of jammy_flows/layers/spheres/sphere_base.py:40-876.

A sphere layer = optional Householder rotation in embedding space (extra_inputs[:, :num_householder_params]) around the layer's
own map, plus the sphere <-> plane chart when it is the first layer of its block.  In-tree layers ('o', 'm', 'f', 'v', 'y') run
all of it in ONE HIP kernel on intrinsic coordinates (``_fused``); third-party subclasses that only provide
``_flow_mapping`` / ``_inv_flow_mapping`` get the rotation and the chart from the 'c' kernels (``inv_flow_mapping`` below).
``num_householder_iter`` doubles as the kernels' rotation code: >= 0 Householder reflections, -1 / -2 / -3 = angles / xyz / quaternion.
"""
import torch
from torch import nn

from .. import layer_base, param_rows
from ... import _hip


class sphere_base(layer_base.layer_base):
    FAMILY = None       # kernel family letter of the concrete layer

    def __init__(self, dimension=1, euclidean_to_sphere_as_first=True, use_permanent_parameters=False, rotation_mode="householder",
                 add_rotation=False, higher_order_cylinder_parametrization=False, num_householder_iter=-1):
        super().__init__(dimension=dimension)
        assert not higher_order_cylinder_parametrization, "higher_order_cylinder_parametrization is disabled (as in the reference)"
        assert dimension in (1, 2), "only S1 and S2 are supported"
        self.higher_order_cylinder_parametrization = False
        self.euclidean_to_sphere_as_first = euclidean_to_sphere_as_first
        self.use_permanent_parameters = use_permanent_parameters
        self.rotation_mode = rotation_mode
        self.add_rotation = add_rotation
        self.num_householder_params = 0
        self.num_householder_iter = 0
        if add_rotation:
            emb = dimension + 1
            if rotation_mode == "householder":
                self.num_householder_iter = emb if num_householder_iter == -1 else num_householder_iter
                self.num_householder_params = self.num_householder_iter * emb
            elif rotation_mode == "angles":                  # Givens rotations (sphere_base.py:132-160)
                self.num_householder_iter = _hip.ROT_CODES["angles"]
                self.num_householder_params = emb * (emb - 1) // 2
            elif rotation_mode in ("xyz", "quaternion"):      # (sphere_base.py:162-216)
                assert dimension == 2
                self.num_householder_iter = _hip.ROT_CODES[rotation_mode]
                self.num_householder_params = 3 if rotation_mode == "xyz" else 4
            else:
                raise Exception("Unknown rotation mode for spheres: ", rotation_mode)
        if use_permanent_parameters and self.num_householder_params > 0:
            self.householder_params = nn.Parameter(torch.randn((1, self.num_householder_params)))
        self.total_param_num += self.num_householder_params
        self._rows = param_rows.PermanentRowCache()

    # ------------------------------------------------------------------------------------------ embedding conversions
    def spherical_to_eucl_embedding(self, x, log_det):
        from ... import autograd
        return autograd.sphere_embedding(x, log_det, self.dimension, True)

    def eucl_to_spherical_embedding(self, x, log_det):
        from ... import autograd
        return autograd.sphere_embedding(x, log_det, self.dimension, False)

    # ------------------------------------------------------------------------------------------ fused path of in-tree layers
    def _layer_tensors(self):
        """permanent tensors AFTER the rotation block, in extra_inputs order (concrete layers override)."""
        return []

    def _permanent_tensors(self):
        """all permanent tensors of the layer in extra_inputs order (rotation block first)"""
        return ([self.householder_params] if self.num_householder_params > 0 else []) + self._layer_tensors()

    def _params_for(self, x, extra_inputs):
        if extra_inputs is None:
            return self._rows.get(self._permanent_tensors(), x, self.total_param_num)
        if extra_inputs.shape[1] != self.total_param_num:
            raise ValueError("extra_inputs has %d columns, layer needs %d" % (extra_inputs.shape[1], self.total_param_num))
        return extra_inputs

    def _fused(self, direction, inputs, extra_inputs, fix_first, **kw):
        x, log_det = inputs
        if direction == "inv":
            first = self.euclidean_to_sphere_as_first if fix_first is None else fix_first
        else:
            first = fix_first if fix_first else self.euclidean_to_sphere_as_first
        first = 1 if first else 0
        emb = bool(self.always_parametrize_in_embedding_space)
        params = self._params_for(x, extra_inputs)
        if direction == "inv":
            if emb:
                x, log_det = self.eucl_to_spherical_embedding(x, log_det if log_det is not None else torch.zeros(x.shape[0], dtype=x.dtype, device=x.device))
            y, ld = _hip.mchain(self.FAMILY, "inv", x, log_det, params, [self.c_struct(first)], self.dimension, **kw)
            if emb and not first:
                y, ld = self.spherical_to_eucl_embedding(y, ld)
            return y, ld
        if emb and not first:
            x, log_det = self.eucl_to_spherical_embedding(x, log_det if log_det is not None else torch.zeros(x.shape[0], dtype=x.dtype, device=x.device))
        y, ld = _hip.mchain(self.FAMILY, "fwd", x, log_det, params, [self.c_struct(first)], self.dimension, **kw)
        if emb:
            y, ld = self.spherical_to_eucl_embedding(y, ld)
        return y, ld

    # ------------------------------------------------------------------------------------------ generic path (third-party subclasses)
    def _c_struct_base(self, hh, first):
        c = _hip.jf_c_layer()
        c.kind, c.hh_iter, c.first, c.lo, c.hi = self.dimension, hh, first, 0.0, 1.0
        return c

    def inv_flow_mapping(self, inputs, extra_inputs=None, include_area_element=True, fix_euclidean_to_sphere_first=None, **kw):
        if self.FAMILY is not None:
            return self._fused("inv", inputs, extra_inputs, fix_euclidean_to_sphere_first, **kw)
        x, log_det = inputs
        emb = bool(self.always_parametrize_in_embedding_space)
        if self.add_rotation:
            rot = self._params_for(x, extra_inputs)[:, :self.num_householder_params]
            if emb:
                x, log_det = self.eucl_to_spherical_embedding(x, log_det)
            x, log_det = _hip.mchain("c", "inv", x, log_det, rot, [self._c_struct_base(self.num_householder_iter, 0)], self.dimension)
            if emb:
                x, log_det = self.spherical_to_eucl_embedding(x, log_det)
        if extra_inputs is None:
            res = self._inv_flow_mapping([x, log_det])
        else:
            res = self._inv_flow_mapping([x, log_det], extra_inputs=extra_inputs[:, self.num_householder_params:],
                                         extra_inputs_base=extra_inputs[:, :self.num_householder_params])
        x, log_det = res[:2]
        first = self.euclidean_to_sphere_as_first if fix_euclidean_to_sphere_first is None else fix_euclidean_to_sphere_first
        if first:
            if emb:
                x, log_det = self.eucl_to_spherical_embedding(x, log_det)
            x, log_det = _hip.mchain("c", "inv", x, log_det, None, [self._c_struct_base(0, 1)], self.dimension)
        return x, log_det

    def flow_mapping(self, inputs, extra_inputs=None, fix_euclidean_to_sphere_first=False, **kw):
        if self.FAMILY is not None:
            return self._fused("fwd", inputs, extra_inputs, fix_euclidean_to_sphere_first, **kw)
        x, log_det = inputs
        emb = bool(self.always_parametrize_in_embedding_space)
        first = fix_euclidean_to_sphere_first if fix_euclidean_to_sphere_first else self.euclidean_to_sphere_as_first
        if first:
            x, log_det = _hip.mchain("c", "fwd", x, log_det, None, [self._c_struct_base(0, 1)], self.dimension)
            if emb:
                x, log_det = self.spherical_to_eucl_embedding(x, log_det)
        if extra_inputs is None:
            x, log_det = self._flow_mapping([x, log_det], sf_extra=None)
        else:
            x, log_det = self._flow_mapping([x, log_det], extra_inputs=extra_inputs[:, self.num_householder_params:],
                                            extra_inputs_base=extra_inputs[:, :self.num_householder_params], sf_extra=None)
        if self.add_rotation:
            rot = self._params_for(x, extra_inputs)[:, :self.num_householder_params]
            if emb:
                x, log_det = self.eucl_to_spherical_embedding(x, log_det)
            x, log_det = _hip.mchain("c", "fwd", x, log_det, rot, [self._c_struct_base(self.num_householder_iter, 0)], self.dimension)
            if emb:
                x, log_det = self.spherical_to_eucl_embedding(x, log_det)
        return x, log_det

    # ------------------------------------------------------------------------------------------ bookkeeping
    def init_params(self, params):
        assert len(params) == self.total_param_num
        if self.add_rotation:
            self.householder_params.data = params[:self.num_householder_params].reshape(1, self.num_householder_params)
            self._init_params(params[self.num_householder_params:])
        else:
            self._init_params(params)

    def get_desired_init_parameters(self):
        parts = []
        if self.num_householder_params > 0:
            # kappa read off the rotation parameters ('f' with kappa_prediction mu / quatvec): start with a tiny kappa (sphere_base.py:716-726)
            small = hasattr(self, "kappa_fn") and self.kappa_fn is None
            parts.append(torch.randn(self.num_householder_params) * (0.01 if small else 1.0))
        parts.append(self._get_desired_init_parameters())
        return torch.cat(parts)

    def obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        if extra_inputs is None:
            self._obtain_layer_param_structure(param_dict, previous_x=previous_x, extra_prefix=extra_prefix)
        else:
            self._obtain_layer_param_structure(param_dict, extra_inputs=extra_inputs[:, self.num_householder_params:], previous_x=previous_x,
                                               extra_prefix=extra_prefix)
        if self.add_rotation:
            param_dict[extra_prefix + "householder"] = extra_inputs[:, :self.num_householder_params] if extra_inputs is not None else self.householder_params

    def _embedding_conditional_return(self, x):
        if x.shape[1] == self.dimension:
            x, _ = _hip.sphere_embedding(x, None, self.dimension, True, want_log_det=False)
        return x

    def _embedding_conditional_return_num(self):
        return self.dimension + 1

    def transform_target_space(self, x, log_det=0.0, transform_from="default", transform_to="embedding"):
        """default / intrinsic / embedding coordinates (sphere_base.py:796-841)."""
        intrinsic_now = True
        if transform_from == "default":
            intrinsic_now = not self.always_parametrize_in_embedding_space
        elif transform_from == "embedding":
            intrinsic_now = False
        assert x.shape[1] == (self.dimension if intrinsic_now else self.dimension + 1)
        want_intrinsic = (transform_to == "intrinsic") or (transform_to == "default" and not self.always_parametrize_in_embedding_space)
        if want_intrinsic == intrinsic_now:
            return x, log_det
        if want_intrinsic:
            return self.eucl_to_spherical_embedding(x, log_det)
        return self.spherical_to_eucl_embedding(x, log_det)

    def _get_layer_base_dimension(self):
        if self.always_parametrize_in_embedding_space and not self.euclidean_to_sphere_as_first:
            return self.dimension + 1
        return self.dimension

    def _init_params(self, params):
        raise NotImplementedError

    def _get_desired_init_parameters(self):
        raise NotImplementedError

    def _inv_flow_mapping(self, inputs, extra_inputs=None, sf_extra=None):
        raise NotImplementedError

    def _flow_mapping(self, inputs, extra_inputs=None, sf_extra=None):
        raise NotImplementedError

    def _obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        raise NotImplementedError
