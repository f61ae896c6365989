"""Drop-in for the reference package `depth_estimation` (MI355X-native)."""
