"""cartpoleplusplus_amd: MI355X-native DDPG-from-pixels training path for cartpole++.

Host orchestration mirrors the reference's Python surface (base_network.Network,
ddpg_cartpole.ActorNetwork / CriticNetwork / DeepDeterministicPolicyGradientAgent,
replay_memory.ReplayMemory / Batch, util.OrnsteinUhlenbeckNoise); everything numerical runs in
hand-written HIP kernels behind the C ABI of include/cartpolepp_abi.h.  No CPU fallback.
"""
__all__ = ["base_network", "ddpg_cartpole", "replay_memory", "util"]
