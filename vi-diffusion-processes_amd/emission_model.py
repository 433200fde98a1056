"""Host-side mirror of markovflow/emission_model.py (`EmissionModel`, emission_model.py:25-153): f_k = H_k x_k."""
import torch


class EmissionModel:
    def __init__(self, emission_matrix, constant_matrix=None):
        """emission_matrix: batch_shape + [T, output_dim, state_dim].  constant_matrix [output_dim, state_dim]: set by callers
        that know H is the same at every time point (stationary kernels, sde_kernel.py:173-211); the Kalman-filter kernels then
        take H as a constant instead of reading a [.., T, o, d] tensor."""
        self.emission_matrix = emission_matrix
        self.output_dim, self.state_dim = emission_matrix.shape[-2], emission_matrix.shape[-1]
        self.constant_matrix = constant_matrix

    def project_state_to_f(self, state):
        """H x (emission_model.py:115-128)."""
        return torch.einsum("...ij,...j->...i", self.emission_matrix, state)

    def project_state_covariance_to_f(self, state_covariance, full_output_cov=True):
        """H S H^T (emission_model.py:130-153)."""
        f = torch.einsum("...ij,...jk,...lk->...il", self.emission_matrix, state_covariance, self.emission_matrix)
        return f if full_output_cov else torch.diagonal(f, dim1=-2, dim2=-1)
