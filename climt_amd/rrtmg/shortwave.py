class RRTMGShortwave:  # placeholder, replaced below in this round
    pass
