"""`fully_amortized_pdf` -- a pdf whose EVERY parameter, including the weights of the inner autoregressive MLPs, is emitted per sample by
one hyper-network (jammy_flows/main/fully_amortized.py:21-330).  Same constructor arguments, attribute names and state_dict layout
(``pdf_to_amortize`` = a ``pdf(..., amortize_everything=True)`` without own parameters, ``amortization_mlp`` = the hyper-network).

Compute on MI355X: the hyper-network is an AmortizableMLP with permanent weights (MFMA dense launches); the inner MLPs receive their
weights row by row and run on ``jf_amlp_stage_*`` (streaming, one wave per sample); the flow layers run on the same chain kernels as the
ordinary pdf, reading their per-sample parameter rows straight out of the hyper-network's output block."""
import numpy
import torch
from torch import nn

from . import default
from ..amortizable_mlp import AmortizableMLP
from ..extra_functions import list_from_str


class fully_amortized_pdf(nn.Module):
    def __init__(self, pdf_defs, flow_defs, options_overwrite=dict(), conditional_input_dim=None, inner_mlp_dims_sub_pdfs="128",
                 inner_mlp_ranks=0, inner_mlp_highway_mode=1, amortization_mlp_dims="128", amortization_mlp_use_custom_mode=True,
                 amortization_mlp_ranks=5, amortization_mlp_highway_mode=0, predict_log_normalization=False, skip_mlp_initialization=False):
        super().__init__()
        assert type(conditional_input_dim) == int, "Fully amortized PDF requires a single encoding with a single dimension!"
        assert predict_log_normalization is False, "TODO: Still need to implement log normalization prediction here."
        self.conditional_input_dim = conditional_input_dim
        self.use_amortizable_mlp = amortization_mlp_use_custom_mode
        self.pdf_to_amortize = default.pdf(pdf_defs, flow_defs, options_overwrite=options_overwrite, conditional_input_dim=None,
                                           amortization_mlp_dims=inner_mlp_dims_sub_pdfs, predict_log_normalization=False,
                                           amortization_mlp_use_custom_mode=True, amortization_mlp_ranks=inner_mlp_ranks,
                                           amortization_mlp_highway_mode=inner_mlp_highway_mode, amortize_everything=True,
                                           skip_mlp_initialization=skip_mlp_initialization)
        for n in ("pdf_defs_list", "flow_defs_list", "total_target_dim", "target_dim_indices_intrinsic", "target_dim_indices_embedded",
                  "target_dim_indices", "base_dim_indices"):
            setattr(self, n, getattr(self.pdf_to_amortize, n))
        hidden = list_from_str(amortization_mlp_dims)
        n_out = self.pdf_to_amortize.total_number_amortizable_params
        if self.use_amortizable_mlp:
            self.amortization_mlp = AmortizableMLP(conditional_input_dim, hidden, n_out, low_rank_approximations=amortization_mlp_ranks,
                                                   use_permanent_parameters=True, highway_mode=amortization_mlp_highway_mode, svd_mode="smart")
            self.total_param_num = self.amortization_mlp.num_amortization_params
        else:
            ins, outs = [conditional_input_dim] + hidden, hidden + [n_out]
            mods, count = [], 0
            for i in range(len(ins)):
                mods.append(nn.Linear(ins[i], outs[i]))
                if i < len(ins) - 1:
                    mods.append(nn.Tanh())
                count += ins[i] * outs[i] + outs[i]
            self.amortization_mlp = default.HipLinearStack(*mods)
            self.total_param_num = count
        self.double()
        if not skip_mlp_initialization:
            self.init_params()

    def forward(self, x, conditional_input=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False):
        """log-probability at x -> (log_prob, log_prob_base, base_pos) (:141-170)"""
        assert conditional_input is not None, "This is by design a conditional PDF .. we require conditional input!"
        all_flow_params = self.amortization_mlp(conditional_input)
        return self.pdf_to_amortize(x, amortization_parameters=all_flow_params, force_embedding_coordinates=force_embedding_coordinates,
                                    force_intrinsic_coordinates=force_intrinsic_coordinates)

    def log_prob(self, x, conditional_input=None, **kwargs):
        return self.forward(x, conditional_input=conditional_input, **kwargs)[0]

    def sample(self, conditional_input=None, samplesize=1, seed=None, allow_gradients=False, force_embedding_coordinates=False,
               force_intrinsic_coordinates=False):
        """(x, base, log_prob, log_prob_base) (:173-213)"""
        assert conditional_input is not None, "This is by design a conditional PDF .. we require conditional input!"
        # the hyper-network runs WITH a graph when gradients are asked for (:173-215: reparameterised training of a fully amortised pdf
        # back-propagates from the samples into amortization_mlp), without one otherwise
        with torch.set_grad_enabled(bool(allow_gradients) and torch.is_grad_enabled()):
            all_flow_params = self.amortization_mlp(conditional_input)
        return self.pdf_to_amortize.sample(amortization_parameters=all_flow_params, seed=seed, allow_gradients=allow_gradients,
                                           force_embedding_coordinates=force_embedding_coordinates,
                                           force_intrinsic_coordinates=force_intrinsic_coordinates)

    def init_params(self, data=None, damping_factor=1000.0, mvn_min_max_sv_ratio=1e-4):
        """(:217-246)"""
        init = self.pdf_to_amortize.init_params(data=data, damping_factor=damping_factor, mvn_min_max_sv_ratio=mvn_min_max_sv_ratio)
        if self.use_amortizable_mlp:
            self.amortization_mlp.initialize_uvbs(fix_final_bias=init, prev_damping_factor=damping_factor)
        else:
            with torch.no_grad():
                for m in self.amortization_mlp:
                    if hasattr(m, "weight"):
                        nn.init.kaiming_uniform_(m.weight.data, a=numpy.sqrt(5))
                        fan_in, _ = nn.init._calculate_fan_in_and_fan_out(m.weight.data)
                        bound = 1 / numpy.sqrt(fan_in)
                        nn.init.uniform_(m.bias.data, -bound, bound)
                        m.weight.data /= damping_factor
                        m.bias.data /= damping_factor
                self.amortization_mlp[-1].bias.data = init.data.to(self.amortization_mlp[-1].bias.data.dtype)

    def count_parameters(self, verbose=False):
        if verbose:
            print("Amoritized PDF param count: \n target PDF pars predicted (not real): %d \n Total PDF (MLP) pars: %d"
                  % (self.pdf_to_amortize.total_number_amortizable_params, self.total_param_num))
        return self.total_param_num
