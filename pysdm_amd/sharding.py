"""placeholder (rewritten below in this round)"""
