import sys; sys.path.insert(0, '.')
import numpy as np
np.random.seed(0)
import bhmm_amd as bhmm
model, observations, states = bhmm.testsystems.generate_synthetic_observations(nstates=3, ntrajectories=8, length=20000)
hmm = bhmm.estimate_hmm(observations, 3)
sampled = bhmm.bayesian_hmm(observations, hmm, nsample=20)
print(hmm.transition_matrix.round(3), sampled.timescales_mean, sampled.timescales_conf)
print(model.transition_matrix.round(3), hmm.output_model.means, type(sampled).__name__)
