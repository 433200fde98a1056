"""
Host-side mirror of markovflow/posterior.py `ConditionalProcess` (posterior.py:166-260): the posterior process
q(s(.)) = int p(s(.) | s(Z)) q(s(Z)) ds(Z) evaluated at arbitrary sorted time points.
"""

from .conditionals import conditional_predict, pairwise_marginals


class ConditionalProcess:
    def __init__(self, posterior_dist, kernel, conditioning_time_points, mean_function=None):
        self.gauss_markov_model = posterior_dist
        self.kernel = kernel
        self.conditioning_time_points = conditioning_time_points
        self.mean_function = mean_function

    def predict_state(self, new_time_points):
        """posterior.py:207-229."""
        pw_mu, pw_cov = pairwise_marginals(self.gauss_markov_model, self.kernel.initial_mean(self.gauss_markov_model.batch_shape),
                                           self.kernel.initial_covariance_matrix())
        return conditional_predict(new_time_points, self.conditioning_time_points, self.kernel, pw_mu, pw_cov)

    def predict_f(self, new_time_points, full_output_cov=False):
        """posterior.py:231-260 (zero mean function unless one is supplied)."""
        em = self.kernel.generate_emission_model(new_time_points)
        m, S = self.predict_state(new_time_points)
        f, fc = em.project_state_to_f(m), em.project_state_covariance_to_f(S, full_output_cov)
        if self.mean_function is not None:
            f = f + self.mean_function(new_time_points)
        return f, fc
