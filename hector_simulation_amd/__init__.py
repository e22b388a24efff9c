"""hector_simulation_amd -- MI355X-native batched force-and-moment MPC QP solver behind HECTOR's convex-MPC C interface."""
__all__ = ["records", "synthetic"]
