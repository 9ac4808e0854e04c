"""MI355X-native batched iLQR for the SE(3) x R^6 quadrotor (hot path of
nitishthatte/QuadrotorILQR).  See DESIGN.md."""
