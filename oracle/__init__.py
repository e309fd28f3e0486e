"""CPU oracle of the reference's FEniCS-backed box heat / elasticity path — test infrastructure only."""
