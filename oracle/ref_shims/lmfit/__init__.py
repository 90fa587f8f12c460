"""Stand-in for the two lmfit helpers the reference's Parameters class imports."""
